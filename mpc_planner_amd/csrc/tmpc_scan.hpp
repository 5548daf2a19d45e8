// Parallel-in-time solve of the interior-point Newton system: the alternative to the sequential square-root Riccati recursion
// (tmpc_riccati.hpp) for callers that buy latency (tmpc_set_latency_mode(h, 2)), one wave per trajectory, all 64 lanes busy.
//
//   min sum_k 1/2 dv_k^T H_k dv_k + g_k^T dv_k    s.t.  dx_{k+1} = F_k dv_k + rb_k  (F = [B A]),  dx_0 = 0
//
// is solved through the block-tridiagonal Schur complement in the dynamics multipliers pi_j (j = 0..N-1: the multiplier of
// dx_{j+1}, dpi_{j+1} of the Riccati path), with P_k = H_k^-1 and E = [0 I]:
//   Y_jj = F_j P_j F_j^T + E P_{j+1} E^T     Y_{j,j-1} = -F_j P_j E^T     beta_j = rb_j - F_j P_j g_j + E P_{j+1} g_{j+1}
//   Y pi = beta  by block cyclic reduction (level l, stride s = 2^l, eliminates the blocks j = s mod 2s),   dv_k = -P_k (g_k + F_k^T pi_k - E^T pi_{k-1})
// Phases and their lane mappings:
//   factor, stage phase      lane = 3 k + s: every lane factorises H_k in registers (7 x 7), the 10 columns of P_k [F_k^T E^T] are dealt
//                            to the stage's three lanes (two triangular solves each), F_k is applied through its structural non-zeros;
//                            the blocks are accumulated in LDS with ds_add_f64 (one wave: program and lane order, deterministic)
//   factor, cyclic reduction one lane per COLUMN of the couplings [Lc Rc] of an eliminated block (two per lane at level 0, ten
//                            blocks): in-register Cholesky of D_j (5 x 5), w = D_j^-1 column, Lc^T w / Rc^T w update the neighbours'
//                            diagonal blocks and create their new coupling; W = D^-1 [Lc Rc] replaces [Lc Rc]
//   solve (per right-hand side)  beta from P_k g_k; forward elimination beta_{j -+ s} -= W^T beta_j and back  (round 6: for the right-hand side the factorisation
//                            saw -- the predictor's -- beta and its forward elimination ride through the stage phase and the reduction levels, whose lanes
//                            hold the columns of W anyway: one dot product and one ds_add_f64 per lane and level; solve(pred) starts at D^-1 beta)
//                            substitution pi_j = D_j^-1 beta_j - W_L pi_{j-s} - W_R pi_{j+s} with one lane per row of a block; the
//                            steps dv_k per stage.  Only g changes between predictor and corrector: the factor is reused.
// dx_0 = 0 and du_N = 0 enter as decoupled blocks of H_0 / H_N (a 1e40 weight on dx_0), so every stage runs the same code.
// Measured against the sequential recursion on real Newton systems of the bench scenes: tools/scan_vs_riccati_bench.hip,
// profiles/round3_f_scan_vs_riccati.json (cycles per system and error against an extended-precision reference).
//
// The [B A] sparsity used here is the unicycle's (rows x, y, psi, v, s; columns a, w, x, y, psi, v, s), exact ones included: the
// hand-written stage functions only (not offered to generated solvers).
#pragma once

// optional clock split (tools/scan_vs_riccati_bench.hip -DTMPC_SCAN_PROFILE; the product never defines it)
#ifdef TMPC_SCAN_PROFILE
__device__ unsigned long long g_scan_clk[16];
#define SCAN_T0() long long sc_t = clock64()
#define SCAN_T(i) do { const long long sc_n = clock64(); if (lane == 0) atomicAdd(&g_scan_clk[i], (unsigned long long)(sc_n - sc_t)); sc_t = clock64(); } while (0)
#else
#define SCAN_T0()
#define SCAN_T(i)
#endif

namespace tmpc {
namespace scan {

constexpr int SNU = 2, SNX = 5, SNV = 7;
// LDS block of one multiplier j: D_j[25] | Lc_j[25] (coupling to j - s; W_L once eliminated) | W_R[25] | beta[5] | chol(D_j)[15].
// The coupling to j + s is not stored: it is the transpose of Lc_{j+s}.
// Every per-lane / per-block stride is an ODD number of doubles: lanes that walk their own copy of a small matrix then hit different
// banks (64-bit accesses; an even stride of 28 doubles put 64 lanes on 8 banks -- measured 2-3x on the phases that stream such arrays).
constexpr int BS = 95, OD = 0, OL = 25, ORR = 50, OB = 75, OLD = 80;
constexpr int LS = 29;                                                      // chol(H_k), packed lower (28, inverse diagonal), per stage
// SL = lanes per stage in the stage phases: 3 (N <= 20: 63 lanes) or 2 (N <= 31: 64 lanes); each lane carries CL of the 11 columns
template <int SL> struct Cfg {
    static_assert(SL == 2 || SL == 3, "lanes per stage");
    static constexpr int CL = (11 + SL - 1) / SL;                           // 4 / 6
    static constexpr int ZL = CL * 7 + ((CL * 7) % 2 == 0 ? 1 : 0);         // per LANE: its columns of P_k [F_k^T E^T g_k] (odd stride)
    static constexpr int NMAX = 64 / SL - 1;                                // 20 / 31
};
constexpr int ZBLK = 36;                                                   // a block of zeros: the coupling of a block without right neighbour (25), [B A] of node N (35)
template <int SL>
__host__ __device__ constexpr int lds_doubles(int N) { return N * BS + (N + 1) * LS + SL * (N + 1) * Cfg<SL>::ZL + (N + 1) * SNV + ZBLK; }

// operands in LDS: Hh packed lower (28 per stage, row-major), BA dense (5 x 7 per stage), gh (7 per stage), rb (5 per stage);
// results dv (7 per stage), dpi (5 per stage, dpi_{j+1} = pi_j); scratch of lds_doubles<SL>(N) at `blk`
template <int SL>
struct ViewT {
    double *Hh; const double *BA, *gh, *rb; double *dv, *dpi, *blk; int N;
    __device__ __forceinline__ double *Ls() const { return blk + N * BS; }
    __device__ __forceinline__ double *Zs() const { return blk + N * BS + (N + 1) * LS; }
    __device__ __forceinline__ double *zg() const { return blk + N * BS + (N + 1) * LS + SL * (N + 1) * Cfg<SL>::ZL; }
    __device__ __forceinline__ double *zeros() const { return zg() + (N + 1) * SNV; }
};
using View = ViewT<3>;

__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

__device__ __forceinline__ double rsqrt3(double d)          // v_rsq_f64 + one third-order step (rsqrt_nr of tmpc_solve.hip)
{
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}
// in-register Cholesky of a packed lower matrix; the diagonal slots receive 1 / L_jj.  Returns true on a non-positive pivot.
template <int n>
__device__ __forceinline__ bool chol_inlane(double (&a)[n * (n + 1) / 2])
{
    bool bad = false;
#pragma unroll
    for (int j = 0; j < n; j++) {
        if (!(a[tri(j, j)] > 0.0)) bad = true;
        const double y = rsqrt3(a[tri(j, j)]);
        a[tri(j, j)] = y;
#pragma unroll
        for (int i = j + 1; i < n; i++) a[tri(i, j)] *= y;
#pragma unroll
        for (int i = j + 1; i < n; i++)
#pragma unroll
            for (int c = j + 1; c <= i; c++) a[tri(i, c)] = fma(-a[tri(i, j)], a[tri(c, j)], a[tri(i, c)]);
    }
    return bad;
}
// R right-hand sides at once, column-oriented (axpy form) and interleaved over the right-hand sides: a lane issues in order, and a
// dependent f64 operation waits ~3 issue slots for its operand -- consecutive instructions here are independent of each other
template <int n, int R>
__device__ __forceinline__ void chol_solve(const double (&a)[n * (n + 1) / 2], double (&v)[R][n])      // v_t <- (L L^T)^-1 v_t
{
#pragma unroll
    for (int c = 0; c < n; c++) {
#pragma unroll
        for (int t = 0; t < R; t++) v[t][c] *= a[tri(c, c)];
#pragma unroll
        for (int i = c + 1; i < n; i++)
#pragma unroll
            for (int t = 0; t < R; t++) v[t][i] = fma(-a[tri(i, c)], v[t][c], v[t][i]);
    }
#pragma unroll
    for (int c = n - 1; c >= 0; c--) {
#pragma unroll
        for (int t = 0; t < R; t++) v[t][c] *= a[tri(c, c)];
#pragma unroll
        for (int i = 0; i < c; i++)
#pragma unroll
            for (int t = 0; t < R; t++) v[t][i] = fma(-a[tri(c, i)], v[t][c], v[t][i]);
    }
}

// structural non-zeros of [B A]: (0,0) (0,1) (0,4) (0,5) (1,0) (1,1) (1,4) (1,5) (2,1) (3,0) (4,0) (4,5); ones at (m, 2 + m)
struct BaRow { double b[12]; };
__device__ __forceinline__ void ba_load(const double *BA, BaRow &r, bool live)
{
    constexpr int at[12] = {0, 1, 4, 5, 7, 8, 11, 12, 15, 21, 28, 33};
#pragma unroll
    for (int e = 0; e < 12; e++) { const double v = BA[at[e]]; r.b[e] = live ? v : 0.0; }      // (load, then select: a conditional load is a branch + a wait)
}
__device__ __forceinline__ void ba_apply(const BaRow &r, const double (&z)[SNV], double (&o)[SNX])         // o = F z
{
    o[0] = fma(r.b[0], z[0], fma(r.b[1], z[1], fma(r.b[2], z[4], fma(r.b[3], z[5], z[2]))));
    o[1] = fma(r.b[4], z[0], fma(r.b[5], z[1], fma(r.b[6], z[4], fma(r.b[7], z[5], z[3]))));
    o[2] = fma(r.b[8], z[1], z[4]);
    o[3] = fma(r.b[9], z[0], z[5]);
    o[4] = fma(r.b[10], z[0], fma(r.b[11], z[5], z[6]));
}
__device__ __forceinline__ void ba_apply_t(const BaRow &r, const double (&p)[SNX], double (&o)[SNV])       // o = F^T p
{
    o[0] = fma(r.b[0], p[0], fma(r.b[4], p[1], fma(r.b[9], p[3], r.b[10] * p[4])));
    o[1] = fma(r.b[1], p[0], fma(r.b[5], p[1], r.b[8] * p[2]));
    o[2] = p[0]; o[3] = p[1];
    o[4] = fma(r.b[2], p[0], fma(r.b[6], p[1], p[2]));
    o[5] = fma(r.b[3], p[0], fma(r.b[7], p[1], fma(r.b[11], p[4], p[3])));
    o[6] = p[4];
}
__device__ __forceinline__ void ba_column(const BaRow &r, int c, double (&v)[SNV])                         // row c of F (column c of F^T)
{
    v[0] = c == 0 ? r.b[0] : c == 1 ? r.b[4] : c == 3 ? r.b[9] : c == 4 ? r.b[10] : 0.0;
    v[1] = c == 0 ? r.b[1] : c == 1 ? r.b[5] : c == 2 ? r.b[8] : 0.0;
    v[2] = c == 0 ? 1.0 : 0.0; v[3] = c == 1 ? 1.0 : 0.0;
    v[4] = c == 0 ? r.b[2] : c == 1 ? r.b[6] : c == 2 ? 1.0 : 0.0;
    v[5] = c == 0 ? r.b[3] : c == 1 ? r.b[7] : c == 3 ? 1.0 : c == 4 ? r.b[11] : 0.0;
    v[6] = c == 4 ? 1.0 : 0.0;
}

// (ds_add_f64: a plain read-modify-write in its place was measured 50 % slower per write phase -- load, wait, add, store)
__device__ __forceinline__ void add_lds(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Loads of a phase are written first and pinned there: left alone, the scheduler sinks every LDS load next to its use, and a lane then
// pays one full LDS round trip per load (s_waitcnt lgkmcnt(0) after each) instead of one per phase.
__device__ __forceinline__ void loads_done() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// H_k of the uniform problem: dx_0 pinned by a decoupled 1e40 block, du_N by a decoupled identity
__device__ __forceinline__ double h_entry(const double *Hk, int k, int N, int i, int j)     // i >= j
{
    const double h = Hk[tri(i, j)];
    if (k == 0 && i >= SNU) return i == j ? 1e40 : 0.0;
    if (k == N && j < SNU) return i == j ? 1.0 : 0.0;
    return h;
}
__device__ __forceinline__ bool g_live(int k, int N, int i) { return !((k == 0 && i >= SNU) || (k == N && i < SNU)); }
// The same entries written INTO Hh once per factorisation (round 6), so that the stage phases load H_k as it is: 25 entries of the state block of
// node 0 (lanes 0 .. 24) and 13 entries of node N touching its inputs (lanes 25 .. 37) instead of two selects on each of the 28 entries in every
// lane (~110 of the stage phase's ~570 instructions).  Hh is rebuilt by the interior-point iteration before every factorisation and nothing
// else reads these entries (dx_0 is fixed, node N has no inputs), so overwriting them changes nothing else.  Followed by a fence / barrier.
template <int SL>
__device__ __forceinline__ void pin_structural_entries(const ViewT<SL> &V, int lane)
{
    if (lane < ZBLK) V.zeros()[lane] = 0.0;
    if (lane < 25) {                                                        // node 0: rows 2 .. 6 of the packed lower triangle = entries 3 .. 27
        const int e = 3 + lane;
        V.Hh[e] = (e == tri(2, 2) || e == tri(3, 3) || e == tri(4, 4) || e == tri(5, 5) || e == tri(6, 6)) ? 1e40 : 0.0;
    } else if (lane < 38) {                                                 // node N: (0,0), (1,0), (1,1), then (i,0), (i,1) for i = 2 .. 6
        const int q = lane - 25;
        const int e = q < 3 ? q : tri(2 + (q - 3) / 2, (q - 3) % 2);
        V.Hh[V.N * 28 + e] = (e == tri(0, 0) || e == tri(1, 1)) ? 1.0 : 0.0;
    }
}

// ---- factor: stage phase ----
// The right-hand side V.gh of the call (the predictor's: complete before the factorisation) rides along as an eleventh column.
template <int SL>
__device__ __forceinline__ bool stage_phase(const ViewT<SL> &V, int lane)
{
    constexpr int CL = Cfg<SL>::CL, ZL = Cfg<SL>::ZL;
    const int N = V.N;
    double *blk = V.blk;
    SCAN_T0();
    pin_structural_entries(V, lane);
    fence();
    const int k = lane / SL, s = lane - SL * k;
    bool bad = false;
    if (lane < SL * (N + 1)) {
        double L[28];
        const double *Hk = V.Hh + k * 28;
#pragma unroll
        for (int e = 0; e < 28; e++) L[e] = Hk[e];
        BaRow F;
        ba_load(k < N ? V.BA + k * SNX * SNV : V.zeros(), F, true);       // (node N has no [B A]: the zero block, no selects)
        double gk[SNV];
#pragma unroll
        for (int i = 0; i < SNV; i++) gk[i] = V.gh[k * SNV + i];
        loads_done();
        bad = chol_inlane<SNV>(L);
        SCAN_T(1);
        if (s == 0) {
#pragma unroll
            for (int e = 0; e < 28; e++) V.Ls()[k * LS + e] = L[e];
        }
        // the lane's CL columns cid = CL s + t of [F^T E^T g] (0..4: F^T, 5..9: E^T, 10: g, 11: none), solved together
        double z[CL][SNV], o[CL][SNX];
#pragma unroll
        for (int t = 0; t < CL; t++) {
            const int cid = CL * s + t;
            double fc[SNV];
            ba_column(F, cid < 5 ? cid : 0, fc);
#pragma unroll
            for (int i = 0; i < SNV; i++) {
                z[t][i] = cid < 5 ? fc[i] : ((i == cid - 5 + SNU) ? 1.0 : 0.0);
                if (t == 10 % CL) z[t][i] = cid == 10 ? (g_live(k, N, i) ? gk[i] : 0.0) : z[t][i];      // (the only slot that can be column 10)
            }
        }
        chol_solve<SNV, CL>(L, z);
#pragma unroll
        for (int t = 0; t < CL; t++) ba_apply(F, z[t], o[t]);
        SCAN_T(2);
        double *Zl = V.Zs() + lane * ZL;
#pragma unroll
        for (int t = 0; t < CL; t++)
#pragma unroll
            for (int i = 0; i < SNV; i++) Zl[t * SNV + i] = z[t][i];
        if (s == 10 / CL) {
#pragma unroll
            for (int i = 0; i < SNV; i++) V.zg()[k * SNV + i] = z[10 % CL][i];
            if (k < N) {                                     // beta_k = rb_k - F_k P_k g_k (stored) ... + E P_{k+1} g_{k+1} (added below: program order of the wave)
#pragma unroll
                for (int i = 0; i < SNX; i++) blk[k * BS + OB + i] = V.rb[k * SNX + i] - o[10 % CL][i];
            }
        }
        // D_k = F P F^T (stored: every column of every block exactly once) ...
#pragma unroll
        for (int t = 0; t < CL; t++) {
            const int cid = CL * s + t;
            if (cid < 5 && k < N) {
#pragma unroll
                for (int i = 0; i < SNX; i++) blk[k * BS + OD + cid * 5 + i] = o[t][i];
            }
        }
        // ... + E P_{k+1} E^T (added: LDS operations of a wave execute in order), and the couplings Y_{k,k-1} = -F_k P_k E^T
#pragma unroll
        for (int t = 0; t < CL; t++) {
            const int cid = CL * s + t, c = cid - 5;
            if (cid >= 5 && cid < 10 && k >= 1) {
#pragma unroll
                for (int i = 0; i < SNX; i++) add_lds(&blk[(k - 1) * BS + OD + c * 5 + i], z[t][SNU + i]);
                if (k < N) {
#pragma unroll
                    for (int i = 0; i < SNX; i++) blk[k * BS + OL + c * 5 + i] = -o[t][i];
                }
            }
        }
        if (s == 10 / CL && k >= 1) {
#pragma unroll
            for (int i = 0; i < SNX; i++) add_lds(&blk[(k - 1) * BS + OB + i], z[10 % CL][SNU + i]);
        }
    }
    fence();
    SCAN_T(3);
    return bad;
}

// ---- factor: one level of cyclic reduction (stride s); CPL columns of [Lc Rc] per lane ----
// MW (four-wave kernels, level 0): the lanes of the level span several waves -- `lane` is the thread index in the workgroup, every wave of the
// workgroup calls (lanes beyond the level's blocks are inactive) and the two fences are workgroup barriers.
template <int CPL, int SL, bool MW = false>
__device__ __forceinline__ bool cr_level(const ViewT<SL> &V, int lane, int s)
{
    constexpr int LPB = (10 + CPL - 1) / CPL;                // lanes per eliminated block (CPL = 3: 4 lanes, column slots 10 and 11 idle)
    const int N = V.N;
    double *blk = V.blk;
    const int m = lane / LPB, q = lane - m * LPB;
    const int o = s * (2 * m + 1), el = o - s, er = o + s;   // eliminated block and its neighbours at this level
    const bool act = o < N, has_r = er < N;
    bool bad = false;
    double w[CPL][SNX], a[CPL][SNX], b[CPL][SNX], wb[CPL];
    double *bo = blk + (act ? o : 0) * BS;
    const double *br = (act && has_r) ? blk + er * BS + OL : V.zeros();   // Rc_o = Lc_er^T (zeros without a right neighbour: no selects on the data)
    SCAN_T0();
    if (act) {
        double L[15];
#pragma unroll
        for (int i = 0; i < SNX; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) L[tri(i, j)] = bo[OD + j * 5 + i];
        double lc[SNX][SNX], rc[SNX][SNX];                   // [i][r] = entry (r, i): column i of the coupling
#pragma unroll
        for (int i = 0; i < SNX; i++)
#pragma unroll
            for (int r = 0; r < SNX; r++) { lc[i][r] = bo[OL + i * 5 + r]; rc[i][r] = br[r * 5 + i]; }
#pragma unroll
        for (int t = 0; t < CPL; t++) {
            const int cid = q * CPL + t, c = cid < 5 ? cid : (cid < 10 ? cid - 5 : 0);
            const double *wp = cid < 5 ? bo + OL + c * 5 : (cid < 10 ? br + c : V.zeros());       // column c of Lc (contiguous) or of Rc = row c of Lc_er (stride 5)
            const int ws = cid < 5 ? 1 : 5;
#pragma unroll
            for (int i = 0; i < SNX; i++) w[t][i] = wp[i * ws];
        }
        double beo[SNX];                                     // beta_o of the right-hand side riding along (complete: every earlier level has fenced)
#pragma unroll
        for (int i = 0; i < SNX; i++) beo[i] = bo[OB + i];
        loads_done();
        SCAN_T(10);
        bad = chol_inlane<SNX>(L);
        SCAN_T(11);
        if (q == 0) {
#pragma unroll
            for (int e = 0; e < 15; e++) bo[OLD + e] = L[e];
        }
        chol_solve<SNX, CPL>(L, w);
#pragma unroll
        for (int t = 0; t < CPL; t++) {                      // (W^T beta_o)_c = w_c . beta_o
            double dsum = 0.0;
#pragma unroll
            for (int r = 0; r < SNX; r++) dsum = fma(w[t][r], beo[r], dsum);
            wb[t] = dsum;
        }
#pragma unroll
        for (int t = 0; t < CPL; t++)
#pragma unroll
            for (int i = 0; i < SNX; i++) { a[t][i] = 0.0; b[t][i] = 0.0; }
#pragma unroll
        for (int r = 0; r < SNX; r++)                        // a = Lc^T w, b = Rc^T w (independent accumulators innermost)
#pragma unroll
            for (int t = 0; t < CPL; t++)
#pragma unroll
                for (int i = 0; i < SNX; i++) { a[t][i] = fma(lc[i][r], w[t][r], a[t][i]); b[t][i] = fma(rc[i][r], w[t][r], b[t][i]); }
    }
    if constexpr (MW) __syncthreads(); else fence();         // every read of the old couplings is done
    SCAN_T(12);
    if constexpr (MW) {
        // several waves: a diagonal block receives one update from its right neighbour's elimination (-Lc^T D^-1 Lc, columns cid < 5) and one from its left
        // neighbour's (-Rc^T D^-1 Rc, cid >= 5), possibly from lanes of different waves -- the two kinds are separated by a barrier, so that every entry sees
        // its two additions in a fixed order (one wave: program order does that)
        static_assert(!MW || CPL == 1, "multi-wave level: one column per lane");
        const int cid = q, c = cid < 5 ? cid : cid - 5;
        if (act) {
#pragma unroll
            for (int i = 0; i < SNX; i++) bo[(cid < 5 ? OL : ORR) + c * 5 + i] = w[0][i];            // W = D^-1 [Lc Rc]
            if (cid < 5) {
#pragma unroll
                for (int i = 0; i < SNX; i++) add_lds(&blk[el * BS + OD + c * 5 + i], -a[0][i]);
                add_lds(&blk[el * BS + OB + c], -wb[0]);                                           // beta_el -= W_L^T beta_o
            }
        }
        __syncthreads();
        if (act && cid >= 5 && has_r) {
#pragma unroll
            for (int i = 0; i < SNX; i++) {
                blk[er * BS + OL + i * 5 + c] = -a[0][i];
                add_lds(&blk[er * BS + OD + c * 5 + i], -b[0][i]);
            }
            add_lds(&blk[er * BS + OB + c], -wb[0]);                                               // beta_er -= W_R^T beta_o
        }
        __syncthreads();
        return bad;
    }
    if (act) {
#pragma unroll
        for (int t = 0; t < CPL; t++) {
            const int cid = q * CPL + t, c = cid < 5 ? cid : cid - 5;
            if (CPL == 3 && cid >= 10) continue;                                               // (idle column slot)
#pragma unroll
            for (int i = 0; i < SNX; i++) bo[(cid < 5 ? OL : ORR) + c * 5 + i] = w[t][i];            // W = D^-1 [Lc Rc]
            if (cid < 5) {
#pragma unroll
                for (int i = 0; i < SNX; i++) add_lds(&blk[el * BS + OD + c * 5 + i], -a[t][i]);       // D_el -= Lc^T D^-1 Lc (whole columns: masking the unused upper triangle costs more than it saves)
                add_lds(&blk[el * BS + OB + c], -wb[t]);                                               // beta_el -= W_L^T beta_o (the right-hand side riding along)
            } else if (has_r) {
#pragma unroll
                for (int i = 0; i < SNX; i++) {
                    blk[er * BS + OL + i * 5 + c] = -a[t][i];                                          // new coupling er -- el: -(Lc^T D^-1 Rc)^T
                    add_lds(&blk[er * BS + OD + c * 5 + i], -b[t][i]);                                 // D_er -= Rc^T D^-1 Rc
                }
                add_lds(&blk[er * BS + OB + c], -wb[t]);                                               // beta_er -= W_R^T beta_o
            }
        }
    }
    if constexpr (MW) __syncthreads(); else fence();
    SCAN_T(13);
    return bad;
}

// cyclic reduction of the blocks the stage phase left in V.blk (per-lane flag: non-positive pivot).  Columns per lane by the number
// of blocks a level eliminates: one where ten lanes per block fit the wave, else two, else three.
template <int SL>
__device__ __forceinline__ bool reduce(const ViewT<SL> &V, int lane)
{
    const int N = V.N;
    bool bad = false;
#ifndef TMPC_SCAN_FORCE_CPL3
    if constexpr (SL == 3) {                                 // N <= 20: ten blocks at level 0 (two columns per lane), at most five afterwards
        bad = cr_level<2>(V, lane, 1);
#pragma unroll 1
        for (int s = 2; s < N; s *= 2) bad |= cr_level<1>(V, lane, s);
    } else
#endif
#pragma unroll 1
    for (int s = 1; s < N; s *= 2) {
        const int ne = (N - s + 2 * s - 1) / (2 * s);        // blocks o = s (2 m + 1) < N
#ifdef TMPC_SCAN_FORCE_CPL3                                  // (tools/scan_vs_riccati_bench.hip: the three-column path on N = 20 systems)
        if (ne * 4 <= 64 && s == 1) { bad |= cr_level<3>(V, lane, s); continue; }
#endif
        if (ne * 10 <= 64) bad |= cr_level<1>(V, lane, s);
        else if (ne * 5 <= 64) bad |= cr_level<2>(V, lane, s);
        else bad |= cr_level<3>(V, lane, s);
    }
    if (lane == 0) {                                         // what is left: block 0
        double L[15];
#pragma unroll
        for (int i = 0; i < SNX; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) L[tri(i, j)] = V.blk[OD + j * 5 + i];
        bad |= chol_inlane<SNX>(L);
#pragma unroll
        for (int e = 0; e < 15; e++) V.blk[OLD + e] = L[e];
    }
    fence();
    return bad;
}
// Factorisation: chol(H_k) -> V.Ls, the reduced blocks -> V.blk.  Returns true (wave-uniform) on a non-positive pivot anywhere.
// (Round 4 measured a second factorisation of the same blocks -- a two-front block Cholesky on two DPP rows, a dozen registers and 45
// doubles per block instead of ~100 and 95 -- correct, but its 10 sequential 5 x 5 steps are a longer chain than the 5 levels here:
// +22 % per tick; profiles/round4_b_parallel_in_time_one_wave_and_twofront.json, HISTORY.md.)
template <int SL>
__device__ __forceinline__ bool factor(const ViewT<SL> &V, int lane)
{
    bool bad = stage_phase(V, lane);
    bad |= reduce(V, lane);
    return __any(bad);
}

// ---- four waves per trajectory (round 6, the control-tick kernel): the factorisation's wide phases on 256 lanes --------------------------------
// Stage phase: ONE column of P_k [F_k^T E^T g_k] per lane instead of four -- wave w holds the stages 5 w .. 5 w + 4 at 11 lanes each (55 lanes); node
// 20 (N = 20 only: it has no F, six columns) sits on the spare lanes 55 .. 60 of wave 3.  Every lane still factorises its stage's H_k (the lanes of
// a wave issue together: redundant lanes cost nothing) but solves and applies F for one column: 56 + 17 instructions where the one-wave phase has
// 224 + 68.  The columns land where the one-wave phase puts them (Z per (stage, column), chol(H_k), P g, the blocks), so reduce() levels >= 1 and
// solve() run unchanged on one wave.  Every block entry is STORED by exactly one lane and then ADDED to by exactly one lane, with a workgroup
// barrier in between: the result does not depend on how the waves interleave.
template <int SL>
__device__ __forceinline__ bool stage_phase4(const ViewT<SL> &V, int tid)
{
    constexpr int CL = Cfg<SL>::CL, ZL = Cfg<SL>::ZL;
    const int N = V.N;
    double *blk = V.blk;
    pin_structural_entries(V, tid);
    __syncthreads();
    const int wv = tid >> 6, l = tid & 63;
    const bool reg = l < 55;
    const int k = reg ? 5 * wv + l / 11 : 20;
    const int cid = reg ? l % 11 : 5 + (l - 55);
    const bool live = reg ? k <= N : (wv == 3 && l < 61 && N == 20);
    const int kc = live ? k : 0;
    bool bad = false;
    double z[1][SNV], o[SNX];
    {
        double L[28];
        const double *Hk = V.Hh + kc * 28;
#pragma unroll
        for (int e = 0; e < 28; e++) L[e] = Hk[e];
        BaRow F;
        ba_load(kc < N ? V.BA + kc * SNX * SNV : V.zeros(), F, true);
        double gk[SNV];
#pragma unroll
        for (int i = 0; i < SNV; i++) gk[i] = V.gh[kc * SNV + i];
        loads_done();
        bad = chol_inlane<SNV>(L) && live;
        if (live && cid == (reg ? 0 : 5)) {                          // one lane per stage keeps chol(H_k)
#pragma unroll
            for (int e = 0; e < 28; e++) V.Ls()[kc * LS + e] = L[e];
        }
        double fc[SNV];
        ba_column(F, cid < 5 ? cid : 0, fc);
#pragma unroll
        for (int i = 0; i < SNV; i++) {
            const double e = (i == cid - 5 + SNU) ? 1.0 : 0.0;
            z[0][i] = cid < 5 ? fc[i] : (cid == 10 ? (g_live(kc, N, i) ? gk[i] : 0.0) : e);
        }
        chol_solve<SNV, 1>(L, z);
        ba_apply(F, z[0], o);
    }
    if (live) {
        double *Zl = V.Zs() + (SL * kc + cid / CL) * ZL + (cid % CL) * SNV;      // (the one-wave phase's layout: lane 3 k + s holds its CL columns)
#pragma unroll
        for (int i = 0; i < SNV; i++) Zl[i] = z[0][i];
        if (cid == 10) {
#pragma unroll
            for (int i = 0; i < SNV; i++) V.zg()[kc * SNV + i] = z[0][i];
            // column slot 11 of the stage does not exist (eleven columns in SL * CL = 12 slots) but solve() multiplies it -- by zero: it has to be FINITE.
            // The one-wave phase computes and stores zeros there; here nobody owns it, so the lane of column 10 clears it (left to whatever the LDS held
            // it turned 17 % of a launch into failures in a process whose earlier kernels had left NaN bit patterns behind: round 6, found by bench.py).
            double *Z11 = V.Zs() + (SL * kc + 11 / CL) * ZL + (11 % CL) * SNV;
#pragma unroll
            for (int i = 0; i < SNV; i++) Z11[i] = 0.0;
        }
        if (cid < 5 && kc < N) {                                              // D_k = F P F^T: stored, every column of every block exactly once
#pragma unroll
            for (int i = 0; i < SNX; i++) blk[kc * BS + OD + cid * 5 + i] = o[i];
        }
        if (cid == 10 && kc < N) {                                            // beta_k = rb_k - F_k P_k g_k: stored here, E P_{k+1} g_{k+1} added after the barrier
#pragma unroll
            for (int i = 0; i < SNX; i++) blk[kc * BS + OB + i] = V.rb[kc * SNX + i] - o[i];
        }
        if (!reg && cid < 10) {                                               // node 20 on the spare lanes: its columns 0 .. 4 (F^T, no F there) are zero --
            double *Z0 = V.Zs() + (SL * kc + (cid - 5) / CL) * ZL + ((cid - 5) % CL) * SNV;      // stored, because solve() multiplies them (by zero)
#pragma unroll
            for (int i = 0; i < SNV; i++) Z0[i] = 0.0;
        }
    }
    __syncthreads();
    if (live && cid >= 5 && cid < 10 && kc >= 1) {                            // ... + E P_{k+1} E^T (added), and the couplings Y_{k,k-1} = -F_k P_k E^T
        const int c = cid - 5;
#pragma unroll
        for (int i = 0; i < SNX; i++) add_lds(&blk[(kc - 1) * BS + OD + c * 5 + i], z[0][SNU + i]);
        if (kc < N) {
#pragma unroll
            for (int i = 0; i < SNX; i++) blk[kc * BS + OL + c * 5 + i] = -o[i];
        }
    }
    if (live && cid == 10 && kc >= 1) {
#pragma unroll
        for (int i = 0; i < SNX; i++) add_lds(&blk[(kc - 1) * BS + OB + i], z[0][SNU + i]);
    }
    __syncthreads();
    return bad;
}

// The same phase for 21 <= N <= 31 (SL = 2: the shipped jackal / jackalsimulator horizon N = 30): TWO columns per lane -- six lanes per stage, ten stages per wave
// (60 lanes), so the 32 nodes fit the four waves without spare-lane tricks.  Lane s6 of a stage takes the columns 2 s6 and 2 s6 + 1; slot 11 is the zero
// column (computed and stored like the one-wave phase does: solve() multiplies it).  Stores, barrier, adds as above.
template <int SL>
__device__ __forceinline__ bool stage_phase4_pairs(const ViewT<SL> &V, int tid)
{
    constexpr int CL = Cfg<SL>::CL, ZL = Cfg<SL>::ZL;
    const int N = V.N;
    double *blk = V.blk;
    pin_structural_entries(V, tid);
    __syncthreads();
    const int wv = tid >> 6, l = tid & 63;
    const int k = 10 * wv + l / 6, s6 = l % 6;
    const bool live = l < 60 && k <= N;
    const int kc = live ? k : 0;
    bool bad = false;
    double z[2][SNV], o[2][SNX];
    {
        double L[28];
        const double *Hk = V.Hh + kc * 28;
#pragma unroll
        for (int e = 0; e < 28; e++) L[e] = Hk[e];
        BaRow F;
        ba_load(kc < N ? V.BA + kc * SNX * SNV : V.zeros(), F, true);
        double gk[SNV];
#pragma unroll
        for (int i = 0; i < SNV; i++) gk[i] = V.gh[kc * SNV + i];
        loads_done();
        bad = chol_inlane<SNV>(L) && live;
        if (live && s6 == 0) {                                                // one lane per stage keeps chol(H_k)
#pragma unroll
            for (int e = 0; e < 28; e++) V.Ls()[kc * LS + e] = L[e];
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int cid = 2 * s6 + t;
            double fc[SNV];
            ba_column(F, cid < 5 ? cid : 0, fc);
#pragma unroll
            for (int i = 0; i < SNV; i++) {
                const double e = (i == cid - 5 + SNU) ? 1.0 : 0.0;            // (cid = 11: i == 6 + SNU never holds -- the zero column)
                z[t][i] = cid < 5 ? fc[i] : (cid == 10 ? (g_live(kc, N, i) ? gk[i] : 0.0) : e);
            }
        }
        chol_solve<SNV, 2>(L, z);
        ba_apply(F, z[0], o[0]);
        ba_apply(F, z[1], o[1]);
    }
    if (live) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int cid = 2 * s6 + t;
            double *Zl = V.Zs() + (SL * kc + cid / CL) * ZL + (cid % CL) * SNV;      // (the one-wave phase's layout)
#pragma unroll
            for (int i = 0; i < SNV; i++) Zl[i] = z[t][i];
            if (cid < 5 && kc < N) {                                          // D_k = F P F^T: stored, every column of every block exactly once
#pragma unroll
                for (int i = 0; i < SNX; i++) blk[kc * BS + OD + cid * 5 + i] = o[t][i];
            }
        }
        if (s6 == 5) {                                                        // column 10 (t = 0): P g, and beta_k = rb_k - F_k P_k g_k (E P_{k+1} g_{k+1} added after the barrier)
#pragma unroll
            for (int i = 0; i < SNV; i++) V.zg()[kc * SNV + i] = z[0][i];
            if (kc < N) {
#pragma unroll
                for (int i = 0; i < SNX; i++) blk[kc * BS + OB + i] = V.rb[kc * SNX + i] - o[0][i];
            }
        }
    }
    __syncthreads();
    if (live && kc >= 1) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int cid = 2 * s6 + t, c = cid - 5;
            if (cid >= 5 && cid < 10) {                                       // ... + E P_{k+1} E^T (added), and the couplings Y_{k,k-1} = -F_k P_k E^T
#pragma unroll
                for (int i = 0; i < SNX; i++) add_lds(&blk[(kc - 1) * BS + OD + c * 5 + i], z[t][SNU + i]);
                if (kc < N) {
#pragma unroll
                    for (int i = 0; i < SNX; i++) blk[kc * BS + OL + c * 5 + i] = -o[t][i];
                }
            }
        }
        if (s6 == 5) {
#pragma unroll
            for (int i = 0; i < SNX; i++) add_lds(&blk[(kc - 1) * BS + OB + i], z[0][SNU + i]);
        }
    }
    __syncthreads();
    return bad;
}

// Four-wave factorisation.  `flag`: four doubles of LDS for the waves' pivot flags.  Level 0 of the cyclic reduction eliminates ten blocks with ten
// coupling columns each: one column per lane on 100 lanes (waves 0 and 1) instead of two per lane on 50; the later levels fit one wave and run there
// while the others wait.  Returns true (workgroup-uniform) on a non-positive pivot anywhere.  SL = 3: N <= 20; SL = 2: N <= 31 (up to 15 blocks at level 0: 150
// lanes on three waves; level 1 has up to eight blocks: two columns per lane on wave 0).
template <int SL>
__device__ __forceinline__ bool factor4(const ViewT<SL> &V, int tid, double *flag)
{
    const int N = V.N;
    bool bad;
    if constexpr (SL == 3) bad = stage_phase4(V, tid); else bad = stage_phase4_pairs(V, tid);
    bad |= cr_level<1, SL, true>(V, tid, 1);
    if (tid < 64) {
#pragma unroll 1
        for (int s = 2; s < N; s *= 2) {
            if (SL == 3 || ((N - s + 2 * s - 1) / (2 * s)) * 10 <= 64) bad |= cr_level<1>(V, tid, s);
            else bad |= cr_level<2>(V, tid, s);
        }
        if (tid == 0) {                                          // what is left: block 0
            double L[15];
#pragma unroll
            for (int i = 0; i < SNX; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) L[tri(i, j)] = V.blk[OD + j * 5 + i];
            bad |= chol_inlane<SNX>(L);
#pragma unroll
            for (int e = 0; e < 15; e++) V.blk[OLD + e] = L[e];
        }
    }
    const bool wbad = __any(bad);
    if ((tid & 63) == 0) flag[tid >> 6] = wbad ? 1.0 : 0.0;
    __syncthreads();
    return (flag[0] != 0.0) | (flag[1] != 0.0) | (flag[2] != 0.0) | (flag[3] != 0.0);
}

// Solve with the factor of the last factor(): V.dv, V.dpi.  pred: the right-hand side is the one factor() saw (its P_k g_k is in
// LDS already); otherwise (V.gh changed since: the corrector) P_k g_k is recomputed, one lane per stage.  V.rb as at factor().
template <int SL>
__device__ __forceinline__ void solve(const ViewT<SL> &V, int lane, bool pred)
{
    constexpr int CL = Cfg<SL>::CL, ZL = Cfg<SL>::ZL;
    const int N = V.N;
    double *blk = V.blk;
    const int k = lane / SL, s3 = lane - SL * k;
    SCAN_T0();
    const bool stage_lane = lane < SL * (N + 1) && s3 == 0;
    if (!pred) {
        for (int e = lane; e < N * SNX; e += 64) blk[(e / SNX) * BS + OB + e % SNX] = V.rb[e];
        if (stage_lane) {
            double L[28], z[1][SNV];
#pragma unroll
            for (int e = 0; e < 28; e++) L[e] = V.Ls()[k * LS + e];
#pragma unroll
            for (int i = 0; i < SNV; i++) { const double gi = V.gh[k * SNV + i]; z[0][i] = g_live(k, N, i) ? gi : 0.0; }
            loads_done();
            chol_solve<SNV, 1>(L, z);
#pragma unroll
            for (int i = 0; i < SNV; i++) V.zg()[k * SNV + i] = z[0][i];
        }
    }
    fence();
    SCAN_T(4);
    if (!pred) {                                             // beta_j = rb_j - F_j P_j g_j + E P_{j+1} g_{j+1}   (pred: the factorisation left beta forward-eliminated)
        const int kc = stage_lane ? k : 0;
        BaRow F;
        ba_load(kc < N ? V.BA + kc * SNX * SNV : V.zeros(), F, true);
        double z[SNV], o[SNX];
#pragma unroll
        for (int i = 0; i < SNV; i++) z[i] = V.zg()[kc * SNV + i];
        loads_done();
        ba_apply(F, z, o);
        if (stage_lane && k < N) {
#pragma unroll
            for (int i = 0; i < SNX; i++) add_lds(&blk[k * BS + OB + i], -o[i]);
        }
        if (stage_lane && k >= 1) {
#pragma unroll
            for (int i = 0; i < SNX; i++) add_lds(&blk[(k - 1) * BS + OB + i], z[SNU + i]);
        }
    }
    fence();
    SCAN_T(5);
    const int m = lane / 5, i5 = lane - 5 * m;              // one lane per row of a block: 12 blocks per pass
#pragma unroll 1
    for (int s = pred ? N : 1; s < N; s *= 2) {              // forward elimination: beta_{o -+ s} -= W^T beta_o
#pragma unroll 1
        for (int m0 = 0; s * (2 * m0 + 1) < N && (SL == 2 || m0 == 0); m0 += 12) {      // (N <= 20: at most ten blocks per level, one pass)
            const int o = s * (2 * (m0 + m) + 1), el = o - s, er = o + s;
            const bool act = o < N && m < 12;
            const double *bo = blk + (act ? o : 0) * BS;     // (clamped: loads are unconditional, only the updates are masked)
            double sa = 0.0, sb = 0.0, be[SNX], wl[SNX], wr[SNX];
#pragma unroll
            for (int r = 0; r < SNX; r++) { be[r] = bo[OB + r]; wl[r] = bo[OL + i5 * 5 + r]; wr[r] = bo[ORR + i5 * 5 + r]; }
            loads_done();
#pragma unroll
            for (int r = 0; r < SNX; r++) { sa = fma(wl[r], be[r], sa); sb = fma(wr[r], be[r], sb); }
            if (act) {
                add_lds(&blk[el * BS + OB + i5], -sa);
                if (er < N) add_lds(&blk[er * BS + OB + i5], -sb);
            }
        }
        fence();
    }
    SCAN_T(6);
    if (lane < N) {                                          // every D_j^-1 beta_j at once
        double *bo = blk + lane * BS;
        double Ld[15], v[1][SNX];
#pragma unroll
        for (int e = 0; e < 15; e++) Ld[e] = bo[OLD + e];
#pragma unroll
        for (int i = 0; i < SNX; i++) v[0][i] = bo[OB + i];
        loads_done();
        chol_solve<SNX, 1>(Ld, v);
#pragma unroll
        for (int i = 0; i < SNX; i++) bo[OB + i] = v[0][i];
    }
    fence();
    SCAN_T(7);
    int s_top = 1;
    while (2 * s_top < N) s_top *= 2;
#pragma unroll 1
    for (int s = s_top; s >= 1; s >>= 1) {                   // back substitution: pi_o = D^-1 beta_o - W_L pi_el - W_R pi_er
#pragma unroll 1
        for (int m0 = 0; s * (2 * m0 + 1) < N && (SL == 2 || m0 == 0); m0 += 12) {
            const int o = s * (2 * (m0 + m) + 1), el = o - s, er = o + s;
            const bool act = o < N && m < 12, has_r = act && er < N;
            const double *bo = blk + (act ? o : 0) * BS, *pl = blk + (act ? el : 0) * BS + OB, *pr = blk + (has_r ? er : 0) * BS + OB;
            double x = bo[OB + i5], y = 0.0, wl[SNX], wr[SNX], xl[SNX], xr[SNX];
#pragma unroll
            for (int r = 0; r < SNX; r++) { wl[r] = bo[OL + r * 5 + i5]; wr[r] = bo[ORR + r * 5 + i5]; xl[r] = pl[r]; xr[r] = pr[r]; }
            loads_done();
#pragma unroll
            for (int r = 0; r < SNX; r++) { x = fma(-wl[r], xl[r], x); y = fma(-wr[r], xr[r], y); }
            if (act) blk[o * BS + OB + i5] = has_r ? x + y : x;  // (a lane reads and replaces its own entry of beta_o only)
        }
        fence();
    }
    SCAN_T(8);
    // dv_k = -P_k g_k - (P_k F_k^T) pi_k + (P_k E^T) pi_{k-1} from the columns the stage phase left in LDS: every lane of a stage sums its own
    {
        const bool lv = lane < SL * (N + 1);
        const int kc = lv ? k : 0;
        const double *Zl = V.Zs() + (lv ? lane : 0) * ZL;
        const double *pk = blk + (kc < N ? kc : 0) * BS + OB, *pm = blk + (kc >= 1 ? kc - 1 : 0) * BS + OB;
        double acc[SNV], pik[SNX], pim[SNX];
#pragma unroll
        for (int i = 0; i < SNX; i++) { pik[i] = pk[i]; pim[i] = pm[i]; }
        double zz[CL * SNV], zgk[SNV];
#pragma unroll
        for (int e = 0; e < CL * SNV; e++) zz[e] = Zl[e];
#pragma unroll
        for (int i = 0; i < SNV; i++) zgk[i] = V.zg()[kc * SNV + i];
        loads_done();
#pragma unroll
        for (int i = 0; i < SNV; i++) acc[i] = s3 == 0 ? -zgk[i] : 0.0;
#pragma unroll
        for (int t = 0; t < CL; t++) {
            const int cid = CL * s3 + t, c = cid < 5 ? cid : cid - 5;
            double a = pik[0], b = pim[0];
#pragma unroll
            for (int cc = 1; cc < SNX; cc++) { a = c == cc ? pik[cc] : a; b = c == cc ? pim[cc] : b; }
            const double coef = cid < 5 ? (kc < N ? -a : 0.0) : ((cid < 10 && kc >= 1) ? b : 0.0);
#pragma unroll
            for (int i = 0; i < SNV; i++) acc[i] = fma(coef, zz[t * SNV + i], acc[i]);
        }
        if (lv && s3 == 0) {
#pragma unroll
            for (int i = 0; i < SNV; i++) V.dv[k * SNV + i] = g_live(k, N, i) ? acc[i] : 0.0;
            if (k < N) {
#pragma unroll
                for (int i = 0; i < SNX; i++) V.dpi[(k + 1) * SNX + i] = pik[i];
            }
        }
        fence();
        if (lv && s3 != 0) {
#pragma unroll
            for (int i = 0; i < SNV; i++) if (g_live(k, N, i)) add_lds(&V.dv[k * SNV + i], acc[i]);
        }
    }
    fence();
    SCAN_T(9);
}

}  // namespace scan
}  // namespace tmpc
