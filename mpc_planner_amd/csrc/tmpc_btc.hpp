// Two-front ("twisted") block Cholesky of the multiplier Schur complement: the alternative to the block cyclic reduction of tmpc_scan.hpp
// for the same system  Y pi = beta  (Y block tridiagonal, SPD, N blocks of 5 x 5:  Y_jj = F_j P_j F_j^T + E P_{j+1} E^T,
// C_j := Y_{j,j-1} = -F_j P_j E^T; built by scan::stage_phase).  Two elimination fronts run at the same time in two 16-lane DPP rows of
// the wave -- the SAME instruction stream, per-lane (base, stride) pairs select the operands:
//   front A (lanes 0..15)   blocks 0, 1, ..., m-1      L_j = chol(D~_j),  W_{j+1} = C_{j+1} L_j^-T,    D~_{j+1} = Y_{j+1,j+1} - W_{j+1} W_{j+1}^T
//   front B (lanes 16..31)  blocks N-1, N-2, ..., m+1  L_j = chol(D^_j),  V_j = C_j^T L_j^-T,          D^_{j-1} = Y_{j-1,j-1} - V_j V_j^T
// and meet at block m = N / 2:  D*_m = Y_mm - W_m W_m^T - V_{m+1} V_{m+1}^T.  Inside a row: lanes 0..4 hold the ROWS of the current 5 x 5
// block, lanes 5..9 the rows of the coupling panel to the next block, which ride through the block's Cholesky like rows below the pivot
// (chol5_rows: the panel trick of the Riccati factorisation's extra row); all cross-lane traffic is v_mov_b64_dpp row_newbcast / row
// shifts, no LDS round trip on the chain.  Against the cyclic reduction: N / 2 sequential 5 x 5 steps instead of log2 N levels, but each
// step is ~130 wave instructions with everything in registers, the blocks need 45 doubles of storage instead of 95 (factor in place, one
// coupling per block), and the register footprint is a dozen doubles per lane instead of ~100 -- what a two-waves-per-SIMD kernel needs.
// The solves are the matching forward / backward substitutions along the two fronts.
#pragma once

namespace tmpc {
namespace scan {

// 5 x 5 panel Cholesky on a 16-lane row: lanes 0..4 = rows of the SPD block (f[c], c <= lane meaningful), lanes 5..9 = rows of the coupling
// panel.  After: lanes 0..4 hold the rows of L (diagonal entry = sqrt of the pivot), rd = 1 / L_ii of the lane's own row; lanes 5..9 the rows
// of  panel L^-T.
__device__ __forceinline__ bool chol5_rows(double (&f)[5], double &rd, int li)
{
    bool bad = false;
    static_for<0, 5>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        const double dpiv = bcast16<c>(f[c]);
        if (!(dpiv > 0.0)) bad = true;
        const double y = rsqrt_nr(dpiv);
        f[c] *= y;
        rd = li == c ? y : rd;
        static_for<c + 1, 5>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const double ljc = bcast16<j>(f[c]);
            f[j] -= f[c] * ljc;
        });
    });
    return bad;
}

// lane i <- lane i + 5 of its row (zero fill): the panel rows (lanes 5..9) handed to the block lanes (0..4)
__device__ __forceinline__ double row_shl5(double x) { return dpp_shift_zero<0x105>(x); }

// Geometry of the two fronts for a lane.
struct Fronts {
    int li, r;               // lane in its row; row index of the lane inside its block / panel (0..4)
    bool A, live, isrow, ispan;
    int m, n_own, n_steps;   // meeting block; blocks this lane's front eliminates; steps of the loop (= front A's count)
    __device__ __forceinline__ Fronts(int lane, int N)
    {
        li = lane & 15;
        const int fr = lane >> 4;
        A = fr == 0; live = fr < 2;
        isrow = li < 5; ispan = li >= 5 && li < 10;
        r = isrow ? li : (ispan ? li - 5 : 0);
        m = N / 2;
        n_steps = m;
        n_own = A ? m : N - 1 - m;
    }
    __device__ __forceinline__ int block(int s, int N) const { return A ? s : N - 1 - s; }               // block eliminated at step s
    __device__ __forceinline__ int next(int s, int N) const { return A ? s + 1 : N - 2 - s; }             // block it updates
    // entry (r, c) of the coupling panel between block(s) and next(s), rows indexed by the NEXT block: A: C_{j+1}[r][c]; B: C_j[c][r]
    __device__ __forceinline__ int pan_base(int s, int N) const { return A ? (s + 1) * BS + OL + r : (N - 1 - s) * BS + OL + r * 5; }
    __device__ __forceinline__ int pan_stride() const { return A ? 5 : 1; }
};

// Factorisation of the blocks scan::stage_phase left in V.blk.  Per block j afterwards: chol (packed lower, reciprocal diagonal) at OLD,
// the coupling to the neighbour towards its front's start replaced by W / V in place.  V.zeros() is the exchange buffer of the meeting block.
template <int SL>
__device__ __forceinline__ bool reduce_twofront(const ViewT<SL> &V, int lane)
{
    const int N = V.N;
    double *blk = V.blk;
    const Fronts g(lane, N);
    double *xch = V.zeros();                                    // (zeroed by stage_phase)
    bool bad = false;
    double f[5], yn[5], pn[5];
    // state entering step s: lanes 0..4 = rows of D~ of block(s), lanes 5..9 = rows of the panel to next(s)
    auto load_block_row = [&](double (&dst)[5], int j) {          // row r of Y_jj (stored column-major and symmetric: entry (r, c) at c * 5 + r)
        const double *p = blk + (j < 0 ? 0 : (j >= N ? N - 1 : j)) * BS + OD + g.r;
#pragma unroll
        for (int c = 0; c < 5; c++) dst[c] = p[c * 5];
    };
    auto load_panel_row = [&](double (&dst)[5], int s) {
        const int sc = s < 0 ? 0 : (s >= N - 1 ? N - 2 : s);       // (clamped: idle steps load something valid and discard it)
        const double *p = blk + g.pan_base(sc, N);
        const int st = g.pan_stride();
#pragma unroll
        for (int c = 0; c < 5; c++) dst[c] = p[c * st];
    };
    {
        double a[5], b[5];
        load_block_row(a, g.block(0, N)); load_panel_row(b, 0);
        loads_done();
#pragma unroll
        for (int c = 0; c < 5; c++) f[c] = g.isrow ? a[c] : b[c];
    }
#pragma unroll 1
    for (int s = 0; s < g.n_steps; s++) {
        const bool act = g.live && s < g.n_own;
        const bool last_own = s == g.n_own - 1;
        // operands of the next step, under this step's Cholesky
        load_block_row(yn, g.next(s, N));
        load_panel_row(pn, s + 1);
        double rd = 0.0;
        const bool b_ = chol5_rows(f, rd, g.li);
        bad |= b_ && act && g.isrow;
        const int j = g.block(s, N);
        if (act && g.isrow) {
            double *Lj = blk + j * BS + OLD;
#pragma unroll
            for (int c = 0; c < 5; c++) if (c <= g.r) Lj[tri(g.r, c)] = c == g.r ? rd : f[c];
        }
        if (act && g.ispan) {
            double *p = blk + g.pan_base(s, N);
            const int st = g.pan_stride();
#pragma unroll
            for (int c = 0; c < 5; c++) p[c * st] = f[c];
        }
        // next block: D~ = Y_next - W W^T, row r in lane r; W's own row comes from lane r + 5
        double w[5], nd[5];
#pragma unroll
        for (int c = 0; c < 5; c++) w[c] = row_shl5(f[c]);
        const bool from_zero = !g.A && last_own;                   // front B's last update goes to the meeting block as a correction only
#pragma unroll
        for (int c = 0; c < 5; c++) nd[c] = from_zero ? 0.0 : yn[c];
        static_for<0, 5>([&](auto c2_) {
            constexpr int c2 = decltype(c2_)::value;
            static_for<0, 5>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                nd[c2] = fma(-w[c], bcast16<5 + c2>(f[c]), nd[c2]);
            });
        });
        if (!g.A && act && last_own && g.isrow) {
#pragma unroll
            for (int c = 0; c < 5; c++) if (c <= g.r) xch[tri(g.r, c)] = nd[c];
        }
#pragma unroll
        for (int c = 0; c < 5; c++) f[c] = g.isrow ? nd[c] : pn[c];
    }
    fence();
    // meeting block: front A's lanes hold Y_mm - W_m W_m^T, front B left -V V^T in the exchange buffer
    {
        double add[5];
#pragma unroll
        for (int c = 0; c < 5; c++) add[c] = xch[tri(g.r, c <= g.r ? c : g.r)];
        loads_done();
#pragma unroll
        for (int c = 0; c < 5; c++) f[c] += add[c];
        double rd = 0.0;
        const bool b_ = chol5_rows(f, rd, g.li);
        bad |= b_ && g.A && g.isrow;
        if (g.A && g.isrow) {
            double *Lj = blk + g.m * BS + OLD;
#pragma unroll
            for (int c = 0; c < 5; c++) if (c <= g.r) Lj[tri(g.r, c)] = c == g.r ? rd : f[c];
        }
    }
    fence();
    return bad;
}

// Forward / backward substitution with the factor of reduce_twofront: beta (at OB of every block) -> pi in place.
template <int SL>
__device__ __forceinline__ void substitute_twofront(const ViewT<SL> &V, int lane)
{
    const int N = V.N;
    double *blk = V.blk;
    const Fronts g(lane, N);
    double *xch = V.zeros();
    struct Ops { double l[5], rd, wrow[5], bnext; };
    auto load_fwd = [&](Ops &o, int s) {                          // L row r of block(s), W row r of the panel to next(s), beta_next[r]
        const int sc = s < 0 ? 0 : (s > N - 2 ? N - 2 : s);
        const int j = g.block(sc, N);
        const double *Lj = blk + j * BS + OLD;
#pragma unroll
        for (int c = 0; c < 5; c++) o.l[c] = Lj[tri(g.r, c < g.r ? c : g.r)];
        o.rd = Lj[tri(g.r, g.r)];
        const double *p = blk + g.pan_base(sc, N);
        const int st = g.pan_stride();
#pragma unroll
        for (int c = 0; c < 5; c++) o.wrow[c] = p[c * st];
        o.bnext = blk[g.next(sc, N) * BS + OB + g.r];
    };
    // ---- forward: y_j = L_j^-1 (beta_j - W_j y_{j-1}) ----
    double t = blk[g.block(0, N) * BS + OB + g.r];
    Ops oa, ob;
    load_fwd(oa, 0);
    loads_done();
    auto fwd_step = [&](const Ops &o, int s) {
        const bool act = g.live && s < g.n_own;
        const bool last_own = s == g.n_own - 1;
        double y[5], yown = 0.0;
        static_for<0, 5>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            y[c] = bcast16<c>(t * o.rd);
            yown = g.li == c ? y[c] : yown;
            t = fma(-o.l[c], y[c], t);                           // (rows r > c; the others are consumed already)
        });
        if (act && g.isrow) blk[g.block(s, N) * BS + OB + g.r] = yown;
        double tn = (!g.A && last_own) ? 0.0 : o.bnext;          // front B's last step: its share of the meeting block's right-hand side only
#pragma unroll
        for (int c = 0; c < 5; c++) tn = fma(-o.wrow[c], y[c], tn);
        if (!g.A && act && last_own && g.isrow) xch[16 + g.r] = tn;
        t = tn;
    };
#pragma unroll 1
    for (int s = 0; s < g.n_steps; s += 2) {
        load_fwd(ob, s + 1);
        fwd_step(oa, s);
        if (s + 1 < g.n_steps) {
            load_fwd(oa, s + 2);
            fwd_step(ob, s + 1);
        }
    }
    fence();
    // ---- meeting block: y_m, then pi_m = L_m^-T y_m (front A's lanes; the result goes to block m's OB for both fronts) ----
    {
        const double *Lm = blk + g.m * BS + OLD;
        double l[5], lc[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { l[c] = Lm[tri(g.r, c < g.r ? c : g.r)]; lc[c] = Lm[tri(c > g.r ? c : g.r, g.r)]; }
        const double rd = Lm[tri(g.r, g.r)];
        const double add = xch[16 + g.r];
        loads_done();
        t += add;
        static_for<0, 5>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            const double yc = bcast16<c>(t * rd);
            t = g.li == c ? yc : (g.li > c ? fma(-l[c], yc, t) : t);      // (rows below the pivot; the rows above hold their y already)
        });
        // t = y_m (lane r holds y_r); back substitution with L_m^T
        static_for<0, 5>([&](auto q_) {
            constexpr int c = 4 - decltype(q_)::value;
            const double xc = bcast16<c>(t * rd);
            t = g.li == c ? xc : (g.li < c ? fma(-lc[c], xc, t) : t);     // (rows r < c: lc[c] = L[c][r]; the rows below hold their x already)
        });
        if (g.A && g.isrow) blk[g.m * BS + OB + g.r] = t;
    }
    fence();
    // ---- backward: pi_j = L_j^-T (y_j - W^T pi_prev), from the meeting block outwards; block(s) for s = n_own - 1 .. 0 ----
    struct OpsB { double lc[5], rd, wcol[5], yj; };
    auto load_bwd = [&](OpsB &o, int s) {
        const int sc = s < 0 ? 0 : (s > N - 2 ? N - 2 : s);
        const int j = g.block(sc, N);
        const double *Lj = blk + j * BS + OLD;
#pragma unroll
        for (int c = 0; c < 5; c++) o.lc[c] = Lj[tri(c > g.r ? c : g.r, g.r)];
        o.rd = Lj[tri(g.r, g.r)];
        // column r of the panel between block(s) and next(s): entries (q, r), q = 0..4
        const double *p = blk + (g.A ? (sc + 1) * BS + OL + g.r * 5 : (N - 1 - sc) * BS + OL + g.r);
        const int st = g.A ? 1 : 5;
#pragma unroll
        for (int q = 0; q < 5; q++) o.wcol[q] = p[q * st];
        o.yj = blk[j * BS + OB + g.r];
    };
    double pi_prev = blk[g.m * BS + OB + g.r];                     // pi_m, both fronts
    OpsB pa, pb;
    load_bwd(pa, g.n_steps - 1);
    loads_done();
    auto bwd_step = [&](const OpsB &o, int s) {
        const bool act = g.live && s < g.n_own;
        double tt = o.yj;
        static_for<0, 5>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            tt = fma(-o.wcol[q], bcast16<q>(pi_prev), tt);
        });
        static_for<0, 5>([&](auto q_) {
            constexpr int c = 4 - decltype(q_)::value;
            const double xc = bcast16<c>(tt * o.rd);
            tt = g.li == c ? xc : (g.li < c ? fma(-o.lc[c], xc, tt) : tt);
        });
        if (act && g.isrow) blk[g.block(s, N) * BS + OB + g.r] = tt;
        pi_prev = act ? tt : pi_prev;                            // (front B idles through front A's extra step when N is even)
    };
#pragma unroll 1
    for (int s = g.n_steps - 1; s >= 0; s -= 2) {
        load_bwd(pb, s - 1);
        bwd_step(pa, s);
        if (s - 1 >= 0) {
            load_bwd(pa, s - 2);
            bwd_step(pb, s - 1);
        }
    }
    fence();
}

}  // namespace scan
}  // namespace tmpc
