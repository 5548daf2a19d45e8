// mpc_planner_amd/csrc/tmpc_fast.hpp -- register-resident interior-point rows (included by tmpc_solve.hip).
//
// Fast variant of the solve kernel for the registered problem shapes.  Differences to the generic kernel:
//   * lane = (stage k, sub-lane c), LPS lanes per stage; every lane OWNS the interior-point rows
//     r = c, c+LPS, c+2 LPS, ... of its stage and keeps their state (t, lam, 1/t, q/t) and their signed
//     coefficients in REGISTERS for the whole QP: the row passes (residuals, barrier Hessian, right-hand sides,
//     step lengths, update) touch LDS only to read the stage's v / dv and to accumulate stage sums with ds_add_f64;
//   * the linearisation's row data [D beta] and the multipliers handed to the next linearisation are staged in LDS
//     that ALIASES the interior-point work arrays (dead at that time), so a trajectory needs ~28 KB instead of 57 KB.
#pragma once

namespace tmpc {

// NLIN >= 0: row counts fixed at compile time (tuned shapes).  NLIN < 0: "runtime shape" -- MM is the number of rows per
// lane the instantiation provides and the actual counts come from Dims (any n_up, M with n_up + M + 14 <= LPS * MM).
template <int NLIN, int MM, int LPS>
struct FastCfg {
    static constexpr bool RT = NLIN < 0;
    static constexpr int NH = RT ? 0 : NLIN + MM;
    static constexpr int NR = NH + 14;                 // general rows + 4 input-box rows + 10 state-box rows
    static constexpr int RPL = RT ? MM : (NR + LPS - 1) / LPS;   // rows per lane
    // Lane c of a stage owns the rows c, c + LPS, c + 2 LPS, ...  For the tuned shapes the KIND of row slot S is mostly a compile-time fact:
    // 0 = a general row (Jacobian in LDS, no box part) for every lane, 1 = a box row (one variable, no Jacobian) for every lane,
    // 2 = depends on the lane (the slot that straddles the boundary) or on run-time row counts.  The row passes skip the part a slot
    // cannot have: a box row loaded the zero triple and multiplied it through, a general row read a dummy variable and multiplied it by 0 --
    // both add exact zeros, so dropping them changes no result bit (signed zeros aside).
    template <int S> static constexpr int KIND = RT ? 2 : ((LPS * S + LPS - 1 < NH) ? 0 : ((LPS * S >= NH) ? 1 : 2));
};

// Row Jacobians (both register-row layouts): stage k's rows start at k * dstride.  Compact layout: dstride = the stage's packed entries + Dims::dpad -- the
// bare stride (40 doubles for cfg 2) puts every fourth stage on the same LDS banks, and the row passes' coefficient loads (five to six passes per
// interior-point iteration) ran into 5-way bank conflicts: +1.0 % at cfg 2 with the padding the host's model picks (tmpc_capi.hip pick_d_pad).  The fast
// layout keeps its bare stride 3 nh: the same padding was measured there too (one wave per SIMD: cfg 3, cfg 5, the ticks) and changed nothing
// (profiles/round5_p_dpad_ab.jsonl).  DPAD_MAX: the range the host searches.
constexpr int DPAD_MAX = 5;
__host__ __device__ inline int lds_doubles_fast(int N, int nh)
{
    const int persistent = (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + (N + 1) * NV + N * NX * NV + N * NX + N * 8 + (N * nh + 1) * 3;
    const int work = (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + 2 * (N + 1) * NV + N * NX + (N + 1) * NV +
                     (N + 1) * NX + (N + 1) * NX + N * NU + 64;
    const int staging = 2 * N * nh;
    return persistent + (work > staging ? work : staging);
}

__device__ __forceinline__ Lds carve_fast(double *s, const Dims &d)
{
    Lds L;
    const int N = d.N;
    L.nh = d.n_up + d.M;
    L.NG = N * L.nh; L.GB = L.NG; L.XB = L.NG + 4 * N; L.nrows = L.XB + 10 * (N - 1);
    auto take = [&](int n) { double *p = s; s += n; return p; };
    L.z = take((N + 1) * NV); L.pi = take((N + 1) * NX); L.W = take((N + 1) * NP28); L.g = take((N + 1) * NV);
    L.BA = take(N * NX * NV); L.b = take(N * NX); L.dyn8 = take(N * 8);
    L.n_pair = 0; L.dstride = 3 * L.nh;                // (the packed layout's addressing with no packed rows; no padding: see DPAD_MAX)
    L.D = take((N * L.nh + 1) * 3);                    // rows' Jacobians stay resident (+ one zero triple for box rows)
    double *w = s;                                      // work region (IPM) ...
    L.v = take((N + 1) * NV); L.pq = take((N + 1) * NX); L.Hh = take((N + 1) * NP28);
    L.rg = take((N + 1) * NV); L.gh = take((N + 1) * NV); L.rb = take(N * NX); L.dv = take((N + 1) * NV);
    L.dpi = take((N + 1) * NX); L.pr = take((N + 1) * NX); L.y = take(N * NU); L.scr = take(64);
    L.scan = s;                                         // (behind the work region: allocated for the parallel-in-time variant only, never touched otherwise)
    s = w;                                              // ... aliased by the staging region (linearisation <-> IPM)
    L.beta = take(N * L.nh); L.lamh = take(N * L.nh);
    L.t = L.lam = L.invt = L.qt = L.rdiag = nullptr;
    return L;
}

// Compact layout (two trajectories' waves per SIMD, eight trajectories per CU: <= 20 KB of LDS and <= 256 registers each).
// LDS keeps only what an interior-point iteration touches at LDS rates: the sparse [B A] table, the packed row Jacobians, the
// QP iterate and the Riccati work arrays.  The NLP-level data -- iterate z, multipliers pi and the stage blocks W, g, b, written
// once per RTI iteration and read once per interior-point iteration -- live in a per-workgroup workspace in global memory
// (one slot per RESIDENT workgroup: ~9 KB x 8 per CU, L2-resident).  L.pr aliases L.dpi (tmpc_riccati.hpp).
// nth = 128: the two-wave instantiations (21 <= N <= 32, four lanes per stage); their block-wide reductions need the 64-entry scratch
// The compact instantiation's Hh layout (the Riccati routines' parameter CP, tmpc_riccati.hpp hstride<CP>): 2 -- blocks at a stride of 29 doubles -- for the tuned
// one-wave shapes; 3 -- 30 doubles, y inside the block, no y array -- for (12,12) (seven per CU with 69 bytes to spare: 29 does not fit, 30 minus the y array does);
// 1 -- the bare 28 -- for the run-time shapes (their LDS size is not known here) and the two-wave instantiations: a wave of theirs holds 16 stages (two-way
// conflicts at most), measured 0 ((20,8), cfg 3) and -2 % ((5,5), the jackal default) with the padded stride (profiles/round5_r_layout_check.jsonl).
// (-DTMPC_EXP_Y_IN_BLOCK, A/B builds of the compact translation unit only: layout 3 for every tuned one-wave shape -- round-5 verdict next-2 (b): for (8,8)
// the 30-double blocks cost 21 doubles and retire the 40-double y array, so the eight-per-CU budget holds; profiles/round6_saturated_levers_ab.jsonl)
#ifdef TMPC_EXP_Y_IN_BLOCK
__host__ __device__ constexpr int compact_layout(int NLIN, int MM, int NTH) { return (NLIN >= 0 && NTH == 64) ? 3 : 1; }
#else
__host__ __device__ constexpr int compact_layout(int NLIN, int MM, int NTH) { return (NLIN >= 0 && NTH == 64) ? ((NLIN == 12 && MM == 12) ? 3 : 2) : 1; }
#endif
__host__ __device__ constexpr int compact_hstride(int layout) { return layout == 2 ? NP28 + 1 : (layout == 3 ? NP28 + 2 : NP28); }
__host__ __device__ inline int lds_doubles_compact(int N, int n_pair, int nh, int nth = 64, int dpad = 0, int layout = 1)
{
    const int dstride = 2 * n_pair + 3 * (nh - n_pair) + dpad;
    const int persistent = N * 8 + BA_NCONST + N * dstride + 3;
    const int work = (N + 1) * NV + (N + 1) * NX + (N + 1) * compact_hstride(layout) + 2 * (N + 1) * NV + N * NX + (N + 1) * NV + (N + 1) * NX + (layout == 3 ? 0 : N * NU) + (nth > 64 ? 64 : 8);
    const int staging = 2 * N * nh;
    return persistent + (work > staging ? work : staging);
}

__device__ __forceinline__ Lds carve_compact(double *s, double *ws, const Dims &d, int nth = 64, int layout = 1)
{
    Lds L;
    const int N = d.N;
    L.nh = d.n_up + d.M;
    L.NG = N * L.nh; L.GB = L.NG; L.XB = L.NG + 4 * N; L.nrows = L.XB + 10 * (N - 1);
    L.n_pair = d.n_lin; L.dstride = 2 * d.n_lin + 3 * (L.nh - d.n_lin) + d.dpad;
    auto take = [&](int n) { double *p = s; s += n; return p; };
    auto takeg = [&](int n) { double *p = ws; ws += n; return p; };
    L.z = takeg((N + 1) * NV); L.pi = takeg((N + 1) * NX); L.W = takeg((N + 1) * NP28); L.g = takeg((N + 1) * NV);
    L.b = takeg((N + 1) * NX);
    L.scan = takeg(N * NP28);                           // (two-wave instantiations: one wave's share of W while the stage is linearised)
    L.BA = nullptr; L.dyn8 = nullptr;
    L.tab = take(N * 8 + BA_NCONST);
    L.D = take(N * L.dstride + 3);                      // (+ one zero triple for box rows / the third entry of topology rows)
    double *w = s;
    L.v = take((N + 1) * NV); L.pq = take((N + 1) * NX); L.Hh = take((N + 1) * compact_hstride(layout));      // (hoff<CP>: tmpc_riccati.hpp)
    L.rg = take((N + 1) * NV); L.gh = take((N + 1) * NV); L.rb = take(N * NX); L.dv = take((N + 1) * NV);
    L.dpi = take((N + 1) * NX); L.pr = L.dpi; L.y = layout == 3 ? nullptr : take(N * NU); L.scr = take(nth > 64 ? 64 : 8);
    s = w;
    L.beta = take(N * L.nh); L.lamh = take(N * L.nh);
    L.t = L.lam = L.invt = L.qt = L.rdiag = nullptr;
    return L;
}

// 1/x: v_rcp_f64 seed + one third-order step (e = 1 - x y; y (1 + e + e^2): error of order e^3, three dependent operations)
__device__ __forceinline__ double rcp_nr(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

__device__ __forceinline__ void lds_add(double *p, double v) { atomicAdd(p, v); }   // ds_add_f64

// ---- Newton-solve policies ------------------------------------------------------------------------------------------------
// How ipm_fast solves the interior-point Newton systems (and synchronises around them).  Solo: the sequential square-root Riccati
// recursion (tmpc_riccati.hpp).  ScanSoloT (below): the parallel-in-time solve (tmpc_scan.hpp).  (Round 3 also had a Team<NQ> policy
// -- several trajectories' sweeps packed into one wave -- measured 4-8 % slower and removed: profiles/round3_b_team_kernels_rejected.json,
// HISTORY.md.)
// SQ: the square-root form of the recursion (tmpc_riccati.hpp; tmpc_dims.riccati_form = TMPC_RICCATI_SQUARE_ROOT) -- SoloSqrt, two run-time-shape instantiations.
template <bool SQ>
struct SoloT {
    static constexpr int NQ = 1;
    __device__ __forceinline__ bool alive() const { return true; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ bool any(bool active) const { return active; }
    template <int NTH, int CP>
    __device__ __forceinline__ bool factor(const Lds &L, const Dims &d, int tl, int sw, int) const { return riccati_factor<NTH, CP, true, SQ>(L, d, tl, sw); }   // (the predictor's backward sweep rides along)
    template <int NTH, int CP>
    __device__ __forceinline__ void solve(const Lds &L, const Dims &d, int tl, int sw, int, int phase, bool) const
    {
        if (phase == 1) riccati_forward<NTH, CP, SQ>(L, d, tl, sw); else riccati_solve<NTH, CP, SQ>(L, d, tl, sw);
    }
};
using Solo = SoloT<false>;
using SoloSqrt = SoloT<true>;

// Latency mode 2: the Newton systems go through the parallel-in-time solve (tmpc_scan.hpp) instead of the Riccati recursion.  Same
// operands in LDS (Hh, [B A], gh, rb), same results (dv, dpi); Hh is left as it is (ipm_fast rebuilds it every iteration anyway).
template <int SL>                                    // lanes per stage of the Newton solve's stage phases: 3 (N <= 20) or 2 (N <= 31)
struct ScanSoloT {
    static constexpr int NQ = 1;
    __device__ __forceinline__ bool alive() const { return true; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ bool any(bool active) const { return active; }
    // NTH = 128 (two waves per trajectory, the row phases on twice the lanes): the Newton solve runs on the wave `sw`, the other waits
    template <int NTH, int CP>
    __device__ __forceinline__ bool factor(const Lds &L, const Dims &d, int tl, int sw, int) const
    {
        static_assert(!CP, "fast layout");
        asm volatile("" : "+v"(tl));                 // opaque per call (see riccati_factor)
        bool bad = false;
        if (NTH == 64 || (tl >> 6) == sw) {
            const scan::ViewT<SL> V{L.Hh, L.BA, L.gh, L.rb, L.dv, L.dpi, L.scan + d.N * NP28, d.N};      // (behind the W shares of the two-wave linearisation)
            bad = scan::factor(V, tl & 63);
            if (NTH > 64 && (tl & 63) == 0) L.scr[63] = bad ? 1.0 : 0.0;
        }
        __syncthreads();
        if (NTH > 64) bad = L.scr[63] != 0.0;
        return bad;
    }
    template <int NTH, int CP>
    __device__ __forceinline__ void solve(const Lds &L, const Dims &d, int tl, int sw, int, int phase, bool) const
    {
        asm volatile("" : "+v"(tl));
        if (NTH == 64 || (tl >> 6) == sw) {
            const scan::ViewT<SL> V{L.Hh, L.BA, L.gh, L.rb, L.dv, L.dpi, L.scan + d.N * NP28, d.N};      // (behind the W shares of the two-wave linearisation)
            scan::solve(V, tl & 63, phase == 1);     // the predictor's right-hand side went through the factorisation
        }
        __syncthreads();
    }
};
using ScanSolo = ScanSoloT<3>;

// Four waves per trajectory (NTH = 256; round 6): the factorisation's wide phases -- the stage phase and level 0 of the cyclic reduction -- run
// on all four waves (scan::factor4), the solves on wave 0 (their levels are a few lanes wide: more waves have nothing to do there).  SL as in ScanSoloT.
template <int SL>
struct ScanQuadT {
    static constexpr int NQ = 1;
    __device__ __forceinline__ bool alive() const { return true; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ bool any(bool active) const { return active; }
    template <int NTH, int CP>
    __device__ __forceinline__ bool factor(const Lds &L, const Dims &d, int tl, int, int) const
    {
        static_assert(!CP && NTH == 256, "fast layout, four waves");
        asm volatile("" : "+v"(tl));
        const scan::ViewT<SL> V{L.Hh, L.BA, L.gh, L.rb, L.dv, L.dpi, L.scan + d.N * NP28, d.N};
        return scan::factor4(V, tl, L.scr + 48);
    }
    template <int NTH, int CP>
    __device__ __forceinline__ void solve(const Lds &L, const Dims &d, int tl, int, int, int phase, bool) const
    {
        asm volatile("" : "+v"(tl));
        if (tl < 64) {
            const scan::ViewT<SL> V{L.Hh, L.BA, L.gh, L.rb, L.dv, L.dpi, L.scan + d.N * NP28, d.N};
            scan::solve(V, tl, phase == 1);
        }
        __syncthreads();
    }
};
using ScanQuad = ScanQuadT<3>;

template <int NLIN, int MM, int LPS, int NTH, int CP, typename PF, typename TEAM = Solo>
__device__ __forceinline__ int ipm_fast(const Lds &L, const Dims &d, int tid, const double *xi, int *iters_out, PF &pf,
                        double (&lam)[FastCfg<NLIN, MM, LPS>::RPL], const TEAM &team = TEAM())
{
    constexpr int NT = NTH;                         // threads per trajectory: 64 (one wave) or 128 (two waves; N > 21)
    using C = FastCfg<NLIN, MM, LPS>;
    constexpr int RPL = C::RPL;
    constexpr bool OCC2 = CP;                       // compact instantiations are built for two waves per SIMD (<= 256 registers)
    // one-wave kernels (three lanes per stage; a stage may straddle two 16-lane rows): the same pre-reduction by two WAVE shifts (DPP wave_shl:1) -- the
    // saturated compact kernel +0.5 % (profiles/round6_one_wave_shift_prereduce_ab.jsonl: three-way same-address ds_add_f64 were 2.6 % of the LDS unit's cycles)
#ifdef TMPC_EXP_NO_WSHL3
    constexpr bool TSUM = false;
#else
    constexpr bool TSUM = LPS == 3 || (LPS == 6 && NTH == 128);      // (six lanes per stage -- the two-wave latency kernels of N <= 21 --: two triples, two adds per entry)
#endif
#ifdef TMPC_EXP_NO_QSUM4
    constexpr bool QSUM = NTH == 256;
#else
    constexpr bool QSUM = NTH == 256 || (NTH == 128 && LPS == 4);      // four-wave kernels (12 or 8 lanes per stage), two-wave kernels at four: the stage sums are pre-reduced per aligned quad
#endif
    static_assert(!QSUM || LPS % 4 == 0, "quad pre-reduction: whole quads per stage");
    // compile-time constants for the tuned shapes, kernel arguments for runtime-shape instantiations
    const int NH = C::RT ? L.nh : C::NH, NR = NH + 14, NLIN_ = C::RT ? d.n_up : NLIN;
    constexpr bool DIET = (LPS == 6 && NLIN == 8) || OCC2;                // 256-register budget (two waves per SIMD): recompute per-row values instead of storing them
    // ... the row steps dt too, unless the instantiation has room for them (round 4: the row-kind specialisation freed ~27 registers of the
    // compact kernels; keeping dt saves three of the five evaluations of c . dv + r_d per row and interior-point iteration)
    constexpr bool DIET_DT = DIET && !(CP && !C::RT && RPL <= 10);
    const int N = d.N;
    // Opaque copy of the lane id: keeps the compiler from hoisting this QP's per-row setup (masks, LDS addresses) out of
    // the RTI loop of the caller, where it would stay live across the register-hungry linearisation and be spilled.
    int tid_q = tid;
    asm volatile("" : "+v"(tid_q));
    // lane -> (stage, sub-lane): a stage never straddles the two waves, so the ds_add_f64 accumulations of a stage all
    // come from one wave in lane order -- results do not depend on how the two waves happen to interleave
    constexpr int SPW = 64 / LPS;                   // stages per wave
    const int wl = tid_q & 63;
    int k = (tid_q >> 6) * SPW + wl / LPS, c = wl % LPS;
    int sl_ = (wl < SPW * LPS && k < N) ? 1 : 0;
    int kk = sl_ ? k : 0;
    // Two-waves-per-SIMD instantiations (256 registers): the lane's indices are made opaque again before every row pass, so that
    // the passes' per-row invariants (LDS addresses, sign constants) are recomputed there instead of being hoisted out of the
    // interior-point loop and kept live -- ~7 registers per row otherwise.
#define ROW_PASS_BEGIN() do { if constexpr (OCC2) asm volatile("" : "+v"(k), "+v"(c), "+v"(kk), "+v"(sl_), "+v"(act), "+v"(box), "+v"(upper), "+v"(varpack)); } while (0)
#define stage_lane (sl_ != 0)
    const double m_rows = (double)(N * NH + 4 * N + 10 * (N - 1));

    // ---- load this lane's rows (signed coefficients) from the staging area into registers ----
    // per row: signed coefficients on (x, y, psi), signed rhs; box rows: sign in `upper`, variable index packed 3 bits/slot
    double sb[RPL];
    // row's Jacobian in L.D (box rows: the zero triple).  Stored unless DIET; the compact tuned shapes with <= 10 rows per lane have the
    // registers to keep the offsets (and those of the third entry of packed rows) since round 4 -- recomputing them cost ~380 instructions per
    // interior-point iteration
    constexpr bool STORE_IDX = !DIET || (CP && !C::RT && RPL <= 10);
    int didx_[STORE_IDX ? RPL : 1];
    constexpr bool STORE_IDX2 = STORE_IDX && CP;
    int didx2_[STORE_IDX2 ? RPL : 1];
    unsigned act = 0, box = 0, upper = 0;
    unsigned long long varpack = 0;
#pragma unroll
    for (int s = 0; s < RPL; s++) {
        const int r = c + LPS * s;
        sb[s] = 0.0;
        if constexpr (STORE_IDX) {
            didx_[s] = N * L.dstride;
            if constexpr (STORE_IDX2) didx2_[s] = N * L.dstride + 2;
        }
        if (stage_lane && r < NR) {
            if (r < NH) {
                const double sgn = (r < NLIN_) ? -1.0 : 1.0;    // topology / slack rows: upper bound 0; ellipsoids: lower bound 1
                if constexpr (STORE_IDX) {
                    if constexpr (CP) {
                        didx_[s] = mul24(k, L.dstride) + (r < L.n_pair ? 2 * r : 3 * r - L.n_pair);
                        if constexpr (STORE_IDX2) didx2_[s] = r >= L.n_pair ? didx_[s] + 2 : N * L.dstride + 2;
                    } else didx_[s] = mul24(k, L.dstride) + 3 * r;
                }
                sb[s] = sgn * L.beta[k * NH + r];
                act |= 1u << s;
            } else {
                const int q = r - NH;                           // 0..3 inputs, 4..13 states
                const int vr = q < 4 ? (q >> 1) : (NU + ((q - 4) >> 1));
                const bool up = q & 1;
                const double sgn = up ? -1.0 : 1.0;
                double bnd = 0.0;
#pragma unroll
                for (int i = 0; i < NV; i++) if (i == vr) bnd = up ? d.ub[i] : d.lb[i];
                box |= 1u << s; if (up) upper |= 1u << s;
                varpack |= (unsigned long long)vr << (3 * s);
                sb[s] = sgn * (bnd - L.z[mul24(k, NV) + vr]);
                if (q < 4 || k >= 1) act |= 1u << s;            // x_0 is fixed, not boxed
            }
        }
    }
    auto VAR = [&](int s) { return (int)((varpack >> (3 * s)) & 7ull); };
    // signed Jacobian of row s (general rows: +-D from LDS; box rows: zero triple)
    // row's Jacobian triple in L.D (box rows and idle lanes: the zero triple behind the last row); recomputed, not stored
    auto DIDX = [&](int s) {
        if constexpr (STORE_IDX) return didx_[STORE_IDX ? s : 0];
        const int r = c + LPS * s;
        if constexpr (CP) return (stage_lane && r < NH) ? mul24(k, L.dstride) + (r < L.n_pair ? 2 * r : 3 * r - L.n_pair) : N * L.dstride;
        return (stage_lane && r < NH) ? mul24(k, L.dstride) + 3 * r : N * L.dstride;
    };
    // third entry of the row's Jacobian: packed rows (topology) have none -- they read the 0.0 of the zero triple, so that the
    // row's arithmetic (sg * 0.0 included) is that of the unpacked layout
    auto DIDX2 = [&](int s) {
        if constexpr (STORE_IDX2) return didx2_[STORE_IDX2 ? s : 0];
        const int r = c + LPS * s;
        if constexpr (CP) return (stage_lane && r < NH && r >= L.n_pair) ? DIDX(s) + 2 : N * L.dstride + 2;
        return DIDX(s) + 2;
    };
#define INVT(s) (LEAN ? rcp_nr(t[s]) : invt_[(LEAN ? 0 : (s))])
    auto CU = [&](int s) { return (box >> s & 1) ? ((upper >> s & 1) ? -1.0 : 1.0) : 0.0; };   // signed unit coefficient
    // box slots of the tuned shapes: with q = r - NH the row's variable is q >> 1 (inputs q = 0..3 -> a, w; states q = 4..13 -> x .. spline) and
    // its side q & 1 -- plain arithmetic on the lane's sub-index instead of the packed masks (which pure shapes then do not carry at all)
    auto VARK = [&](auto s_) {
        constexpr int s = decltype(s_)::value;
        if constexpr (C::template KIND<s> == 1) return (c + LPS * s - C::NH) >> 1;
        else return VAR(s);
    };
    auto CUK = [&](auto s_) {
        constexpr int s = decltype(s_)::value;
        if constexpr (C::template KIND<s> == 1) return ((c + LPS * s - C::NH) & 1) ? -1.0 : 1.0;
        else return CU(s);
    };
    // "this slot holds a live row": a general slot of a tuned shape is live on every stage lane, a box slot whose rows are all state boxes on
    // every stage lane but stage 0 (x_0 is fixed, not boxed) -- one comparison per pass instead of a bit test per slot and pass
    auto ACT = [&](auto s_) {
        constexpr int s = decltype(s_)::value;
        constexpr int K = C::template KIND<s>;
        if constexpr (K == 0) return sl_ != 0;
        else if constexpr (K == 1 && LPS * s - C::NH >= 4 && LPS * s + LPS - 1 < C::NR) return sl_ != 0 && k >= 1;
        else return (act >> s & 1) != 0;
    };
    // SIGNED Jacobian of row slot s: the linearisation's sink stores sgn * D (fast layouts), box slots have none
    auto coef = [&](auto s_, double &c0, double &c1, double &c2) {
        constexpr int s = decltype(s_)::value;
        if constexpr (C::template KIND<s> == 1) { c0 = 0.0; c1 = 0.0; c2 = 0.0; }
        else {
            const double *Dr_ = L.D + DIDX(s); c0 = Dr_[0]; c1 = Dr_[1];
#ifdef TMPC_EXP_C2_LITERAL
            // A/B (round 6): a slot whose rows are all packed topology rows has no third entry -- the load of the zero triple's 0.0 is an LDS instruction per slot
            // and pass; the literal is the same value (valid only while n_lin == NLIN: the tuned shapes without scenario / decomp rows)
            if constexpr (CP && !C::RT && C::template KIND<s> == 0 && LPS * s + LPS - 1 < NLIN) c2 = 0.0; else
#endif
            c2 = L.D[DIDX2(s)];
        }
    };
    // c . w for row slot s and a stage vector w = (.., x, y, p at ZX, ZY, ZPSI ..): Jacobian part + box part, whichever the slot can have
    auto rowdot = [&](auto s_, double c0, double c1, double c2, double x, double y, double pp, const double *vec) {
        constexpr int s = decltype(s_)::value;
        constexpr int K = C::template KIND<s>;
        if constexpr (K == 0) return c0 * x + c1 * y + c2 * pp;
        else if constexpr (K == 1) {
            // box slot of a tuned shape: q = r - NH, variable q >> 1, side q & 1: +-w[var] by flipping the sign bit (exactly (+-1.0) * w[var])
            // the LAST slot of a tuned shape can run past the stage's rows (r >= NR on its upper sub-lanes: q >> 1 would be NV, the next node's
            // first entry): the index is clamped there -- one v_min on that one slot -- so that nothing outside the stage is read even though
            // every consumer masks such a row with ACT() (invariant: a value returned for a row with ACT() == false is never used unmasked)
            int q = c + LPS * s - C::NH;
            if constexpr (LPS * s + LPS - 1 >= C::NR) q = q < 2 * NV - 1 ? q : 2 * NV - 1;
            const unsigned long long u = __builtin_bit_cast(unsigned long long, vec[q >> 1]) ^ ((unsigned long long)((unsigned)q << 31) << 32);
            return __builtin_bit_cast(double, u);
        }
        else return c0 * x + c1 * y + c2 * pp + CU(s) * vec[VAR(s)];
    };
    team.sync();                                             // staging is dead from here on
    // QP start: dz = 0 except dx_0 = xinit - x_0; duals 0
    for (int e = tid_q; e < (N + 1) * NV; e += NT) L.v[e] = 0.0;
    for (int e = tid_q; e < (N + 1) * NX; e += NT) L.pq[e] = 0.0;
    team.sync();
    if (tid_q < NX) L.v[NU + tid_q] = xi[tid_q] - L.z[NU + tid_q];
    team.sync();

    ROW_PASS_BEGIN();
    double t[RPL], qt[RPL];
    constexpr bool LEAN = RPL > 10 || (LPS == 6 && NLIN == 8) || OCC2;    // recompute 1/t instead of keeping it: many rows per lane, or the 256-register
                                                    // budget of the two-waves-per-SIMD instantiations
    double invt_[LEAN ? 1 : RPL];               // the row residual r_d = c.v - sb - t is recomputed where needed
    {
        const double vx = L.v[mul24(kk, NV) + ZX], vy = L.v[mul24(kk, NV) + ZY], vp = L.v[mul24(kk, NV) + ZPSI];
        static_for<0, RPL>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            double c0s, c1s, c2s;
            coef(s_, c0s, c1s, c2s);
            const double r0 = rowdot(s_, c0s, c1s, c2s, vx, vy, vp, L.v + mul24(kk, NV)) - sb[s];
            t[s] = r0 > d.thr0 ? r0 : d.thr0;
            if constexpr (!LEAN) invt_[s] = rcp_nr(t[s]);
            lam[s] = ACT(s_) ? d.mu0 / t[s] : 0.0;
            qt[s] = 0.0;
        });
    }
    // `active`: this trajectory's QP is still iterating.  One trajectory per workgroup (Solo): the loop ends with it.  Teams: the
    // trajectories of a workgroup iterate in lock step (the Riccati sweeps of all of them run in one wave), so a finished one
    // keeps attending the team's barriers until every member is done.
    int status = 2, iters = 0;
    bool active = team.alive();
    for (int it = 0;; it++) {
        pf.start();
        int tl = tid;                                    // per-iteration opaque lane id: per-lane addresses of the item loops and
        asm volatile("" : "+v"(tl));                   // sweeps are recomputed, not kept live (and spilled) across the whole solve
        double res_d = 0.0, res_m = 0.0, mu = 0.0;
        if (active) {
        // ---- stage parts of the residuals: rg0 = g + W v + [B A]^T pi_{k+1} - [0; pi_k];  rb;  Hh <- W ----
        double res_b = 0.0;
        if constexpr (!CP) {
            // Fast layout (one wave per SIMD: nothing else hides an LDS round trip): every operand of an element is loaded unconditionally
            // (clamped indices) ahead of the arithmetic and pinned there -- a `cond ? lds[i] : 0` is a branch and a wait per load, and
            // the scheduler sinks unpinned loads next to their uses.  Same sums in the same order as the compact form below.
            for (int e = tl; e < (N + 1) * NV; e += NT) {
                const int ks = e / NV, i = e - mul24(ks, NV);
                const bool skip = (ks == N && i < NU) || (ks == 0 && i >= NU);
                const int kb = ks < N ? ks : N - 1;
                const double *Wk = L.W + mul24(ks, NP28), *vk = L.v + mul24(ks, NV), *BA = L.BA + mul24(kb, NX) * NV, *pn = L.pq + mul24(kb + 1, NX);
                double ge = L.g[e], wv[NV], vv[NV], bav[NX], pv[NX];
#pragma unroll
                for (int j = 0; j < NV; j++) { wv[j] = Wk[sidx(i, j)]; vv[j] = vk[j]; }
#pragma unroll
                for (int l = 0; l < NX; l++) { bav[l] = BA[l * NV + i]; pv[l] = pn[l]; }
                const double pm = L.pq[mul24(ks >= 1 ? ks : 1, NX) + (i >= NU ? i - NU : 0)];
                scan::loads_done();
                double acc = ge;
#pragma unroll
                for (int j = 0; j < NV; j++) acc += wv[j] * vv[j];
                if (ks < N) {
#pragma unroll
                    for (int l = 0; l < NX; l++) acc += bav[l] * pv[l];
                }
                if (i >= NU && ks >= 1) acc -= pm;
                acc = skip ? 0.0 : acc;
                L.rg[e] = acc; L.gh[e] = acc;
            }
            for (int e = tl; e < N * NX; e += NT) {
                const int ks = e / NX, i = e - mul24(ks, NX);
                const double *vk = L.v + mul24(ks, NV), *BA = L.BA + mul24(ks, NX) * NV + i * NV;
                double be = L.b[e], vn = L.v[(ks + 1) * NV + NU + i], bav[NV], vv[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) { bav[j] = BA[j]; vv[j] = vk[j]; }
                scan::loads_done();
                double acc = be - vn;
#pragma unroll
                for (int j = 0; j < NV; j++) acc += bav[j] * vv[j];
                L.rb[e] = acc;
                res_b = fmax(res_b, fabs(acc));
            }
        } else {
            // Compact layout (round 4): lane = NODE ks.  The node's W block comes straight from the global workspace into registers (and from
            // there into Hh: no separate copy pass and no barrier before the residuals), and with the node fixed per lane every index of the
            // residuals -- the packed-W entries, the structural non-zeros of [B A] (dyn8 entries, 1, dt, dt^2/2) -- is a compile-time constant:
            // ~115 instructions where the element-parallel form (an integer division, seven symmetric-index computations and five / seven table
            // decodes per element, three passes) spent ~700.  Sums in the same order as before; structural zeros of [B A] are skipped (they added
            // exact zeros), its ones are additions.
            const bool nd = tl <= N;
            const int ks = nd ? tl : N;                    // (idle lanes redo node N and store nothing)
            const int kb = ks < N ? ks : N - 1;            // stage whose [B A], pi_{k+1}, b the node reads
            double w[NP28], vk[NV], pn[NX], po[NX], d8[8], gk[NV], bk[NX], vn[NX];
            {
                const double *Wg = L.W + mul24(ks, NP28);
#pragma unroll
                for (int e = 0; e < NP28; e++) w[e] = Wg[e];
#pragma unroll
                for (int j = 0; j < NV; j++) { vk[j] = L.v[mul24(ks, NV) + j]; gk[j] = L.g[mul24(ks, NV) + j]; }
#pragma unroll
                for (int l = 0; l < NX; l++) {
                    pn[l] = L.pq[mul24(kb + 1, NX) + l]; po[l] = L.pq[mul24(ks >= 1 ? ks : 1, NX) + l];
                    bk[l] = L.b[mul24(kb, NX) + l]; vn[l] = L.v[mul24(kb + 1, NV) + NU + l];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) d8[q] = L.tab[kb * 8 + q];
            }
            if (nd) {
#pragma unroll
                for (int e = 0; e < NP28; e++) L.Hh[hoff_lane<CP>(ks) + e] = w[e];
            }
            const double dtc = d.dt, sdtc = d.sdt, shc = d.shdt2;      // (spline row of [B A]: zero for the model without a spline state)
            const double Xa = d8[D8_XA], Xw = d8[D8_XW], Xp = d8[D8_XP], Xv = d8[D8_XV], Ya = d8[D8_YA], Yw = d8[D8_YW], Yp = d8[D8_YP], Yv = d8[D8_YV];
            double rgk[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) {
                double acc = gk[i];
#pragma unroll
                for (int j = 0; j < NV; j++) acc += w[sidx(i, j)] * vk[j];
                rgk[i] = acc;
            }
            if (ks < N) {                                  // + [B A]^T pi_{k+1}, column by column
                rgk[ZA] = ((rgk[ZA] + Xa * pn[0]) + Ya * pn[1] + dtc * pn[3]) + shc * pn[4];
                rgk[ZW] = (rgk[ZW] + Xw * pn[0]) + Yw * pn[1] + dtc * pn[2];
                rgk[ZX] += pn[0];
                rgk[ZY] += pn[1];
                rgk[ZPSI] = ((rgk[ZPSI] + Xp * pn[0]) + Yp * pn[1]) + pn[2];
                rgk[ZV] = (((rgk[ZV] + Xv * pn[0]) + Yv * pn[1]) + pn[3]) + sdtc * pn[4];
                rgk[ZS] += pn[4];
            }
#pragma unroll
            for (int i = 0; i < NV; i++) {
                double acc = rgk[i];
                if (i >= NU && ks >= 1) acc -= po[i - NU];
                const bool skip = (ks == N && i < NU) || (ks == 0 && i >= NU);
                acc = skip ? 0.0 : acc;
                if (nd) { L.rg[mul24(ks, NV) + i] = acc; L.gh[mul24(ks, NV) + i] = acc; }
            }
            {                                              // rb = b + [B A] v_k - dx_{k+1}, row by row
                double rbk[NX];
                rbk[0] = (((((bk[0] - vn[0]) + Xa * vk[ZA]) + Xw * vk[ZW]) + vk[ZX]) + Xp * vk[ZPSI]) + Xv * vk[ZV];
                rbk[1] = (((((bk[1] - vn[1]) + Ya * vk[ZA]) + Yw * vk[ZW]) + vk[ZY]) + Yp * vk[ZPSI]) + Yv * vk[ZV];
                rbk[2] = ((bk[2] - vn[2]) + dtc * vk[ZW]) + vk[ZPSI];
                rbk[3] = ((bk[3] - vn[3]) + dtc * vk[ZA]) + vk[ZV];
                rbk[4] = (((bk[4] - vn[4]) + shc * vk[ZA]) + sdtc * vk[ZV]) + vk[ZS];
                if (tl < N) {
#pragma unroll
                    for (int i = 0; i < NX; i++) { L.rb[mul24(ks, NX) + i] = rbk[i]; res_b = fmax(res_b, fabs(rbk[i])); }
                }
            }
        }
        if constexpr (!CP) for (int e = tl; e < (N + 1) * NP28; e += NT) L.Hh[e] = L.W[e];
        team.sync();
        pf.stop(PH_HH);                                   // (profile: stage-vector part of the residuals)
        // ---- row pass R (registers): residuals, rg -= lam c, Hh += d c c^T, gh += d rd c ----
        ROW_PASS_BEGIN();
        {
            const double vx = L.v[mul24(kk, NV) + ZX], vy = L.v[mul24(kk, NV) + ZY], vp = L.v[mul24(kk, NV) + ZPSI];
            double gs0 = 0, gs1 = 0, gs2 = 0, rs0 = 0, rs1 = 0, rs2 = 0;
            double h00 = 0, h10 = 0, h11 = 0, h20 = 0, h21 = 0, h22 = 0;
            static_for<0, RPL>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                constexpr int K = C::template KIND<s>;
                const bool a = ACT(s_);
                double c0s, c1s, c2s;
                coef(s_, c0s, c1s, c2s);
                const double r = rowdot(s_, c0s, c1s, c2s, vx, vy, vp, L.v + mul24(kk, NV)) - sb[s] - t[s];
                const double rds = a ? r : 0.0;
                const double comp = lam[s] * t[s];
                const double dd = lam[s] * INVT(s);
                const double w = dd * rds;
                if (a) { res_d = fmax(res_d, fabs(r)); res_m = fmax(res_m, comp); mu += comp; }
                if constexpr (K != 1) {
                    gs0 += lam[s] * c0s; gs1 += lam[s] * c1s; gs2 += lam[s] * c2s;
                    rs0 += w * c0s; rs1 += w * c1s; rs2 += w * c2s;
                    const double d0 = dd * c0s, d1 = dd * c1s, d2 = dd * c2s;
                    h00 += d0 * c0s; h10 += d1 * c0s; h11 += d1 * c1s;
                    h20 += d2 * c0s; h21 += d2 * c1s; h22 += d2 * c2s;
                }
                if constexpr (K != 0) {
                    if (a && (K == 1 || (box >> s & 1))) {                // box row: one variable
                        const int vr = VARK(s_);
                        const double cus = CUK(s_);
                        lds_add(&L.rg[mul24(k, NV) + vr], -lam[s] * cus);
                        lds_add(&L.gh[mul24(k, NV) + vr], w * cus);
                        lds_add(&L.Hh[hoff_lane<CP>(k) + pidx(vr, vr)], dd);
                    }
                }
            });
            // Twelve lanes per stage (the four-wave kernel): a ds_add_f64 whose lanes hit the same address is serialised, and twelve lanes of a stage adding to
            // one entry cost that phase twice what six did (measured: 328 k instead of 164 k cycles per solve).  The stage's lanes 12 s .. 12 s + 11 are three
            // aligned quads: two DPP quad permutations sum each quad in registers, its first lane adds -- three-way conflicts, like the one-wave kernels.
            if constexpr (QSUM) {
                // (eight lanes per stage: a stage is half a 16-lane row, so one row shift by four folds its second quad into the first: no conflict left)
                auto quad = [](double x) { x += dpp_move<0xB1, 0xf>(x, 0.0); x += dpp_move<0x4E, 0xf>(x, 0.0); if constexpr (LPS == 8) x += dpp_shift_zero<0x104>(x); return x; };      // quad_perm [1,0,3,2], then [2,3,0,1]; row_shl:4
                gs0 = quad(gs0); gs1 = quad(gs1); gs2 = quad(gs2); rs0 = quad(rs0); rs1 = quad(rs1); rs2 = quad(rs2);
                h00 = quad(h00); h10 = quad(h10); h11 = quad(h11); h20 = quad(h20); h21 = quad(h21); h22 = quad(h22);
            }
            if constexpr (TSUM) {                 // three lanes per stage: x[l] + x[l + 1] + x[l + 2] by two wave shifts (wave_shl:1), the stage's first lane adds
                auto tri3 = [](double x) { const double t = x + dpp_shift_zero<0x130>(x); return x + dpp_shift_zero<0x130>(t); };
                gs0 = tri3(gs0); gs1 = tri3(gs1); gs2 = tri3(gs2); rs0 = tri3(rs0); rs1 = tri3(rs1); rs2 = tri3(rs2);
                h00 = tri3(h00); h10 = tri3(h10); h11 = tri3(h11); h20 = tri3(h20); h21 = tri3(h21); h22 = tri3(h22);
            }
            if (stage_lane && (!QSUM || (c & (LPS == 8 ? 7 : 3)) == 0) && (!TSUM || c == 0 || c == 3)) {
                lds_add(&L.rg[mul24(k, NV) + ZX], -gs0); lds_add(&L.rg[mul24(k, NV) + ZY], -gs1); lds_add(&L.rg[mul24(k, NV) + ZPSI], -gs2);
                lds_add(&L.gh[mul24(k, NV) + ZX], rs0); lds_add(&L.gh[mul24(k, NV) + ZY], rs1); lds_add(&L.gh[mul24(k, NV) + ZPSI], rs2);
                double *Hk = L.Hh + hoff_lane<CP>(k);
                lds_add(&Hk[pidx(ZX, ZX)], h00); lds_add(&Hk[pidx(ZY, ZX)], h10); lds_add(&Hk[pidx(ZY, ZY)], h11);
                lds_add(&Hk[pidx(ZPSI, ZX)], h20); lds_add(&Hk[pidx(ZPSI, ZY)], h21); lds_add(&Hk[pidx(ZPSI, ZPSI)], h22);
            }
        }
        team.sync();
        // now rg = rg0 - sum lam c (residual) and gh = rg0 + sum d rd c (predictor rhs, q/t = lam)
        double res_g = 0.0;
        for (int e = tl; e < (N + 1) * NV; e += NT) {
            const double r = L.rg[e];                                          // (load, then select: see the residual loops above)
            res_g = fmax(res_g, (e >= NU && e < NV) ? 0.0 : fabs(r));          // dx_0 is fixed: its stationarity row is not a residual
        }
        double res_worst;
        blk_residuals<NTH>(res_worst, res_g, res_b, res_d, res_m, mu, L.scr, tl);
        mu = mu / m_rows;
        pf.stop(PH_RES);
        if (!isfinite(res_worst)) { status = 4; active = false; }
        else if (res_worst <= d.qp_tol) { status = 0; active = false; }
        else if (it >= d.qp_iter_max) { status = 2; active = false; }
        else iters = it + 1;
        }
        if (!team.any(active)) break;                    // (teams: one barrier; the votes also tell the sweeping wave which rows are live)

        // two-wave variant: the sweeping wave alternates with the iteration (and differs between neighbouring trajectories),
        // so that co-resident trajectories seldom run the same sweep on the same SIMD
        const int sw = NTH >= 128 ? ((it + (int)(blockIdx.x >> 8)) & 1) : 0;
        const bool fbad = team.template factor<NTH, CP>(L, d, tl, sw, it);
        pf.stop(PH_FACTOR);
        if (active && fbad) { status = 4; active = false; }
        if (TEAM::NQ == 1 && !active) break;
        // ---- predictor: rhs = rg + sum c (lam + d rd)  (q/t = lam) ----
        team.template solve<NTH, CP>(L, d, tl, 1 - sw, it, 1, active);
        pf.stop(PH_SOLVE);
        // Row step dt = c.dv + r_d is recomputed from the direction in LDS wherever it is needed (no per-row storage:
        // the register budget decides how many waves a SIMD holds)
        auto row_dt = [&](auto s_, double dx, double dy, double dp, double vx, double vy, double vp) {
            constexpr int s = decltype(s_)::value;
            const bool a = ACT(s_);
            double c0s, c1s, c2s;
            coef(s_, c0s, c1s, c2s);
            const double rds = rowdot(s_, c0s, c1s, c2s, vx, vy, vp, L.v + mul24(kk, NV)) - sb[s] - t[s];
            const double ddot = rowdot(s_, c0s, c1s, c2s, dx, dy, dp, L.dv + mul24(kk, NV));
            return a ? ddot + rds : 0.0;
        };
        double dt_[DIET_DT ? 1 : RPL];                   // row steps (stored unless DIET_DT)
        // Step lengths: alpha_max = min over rows of -t/dt (dt < 0) and -lam/dl (dl < 0) is taken as 1 / max of the reciprocal
        // ratios -dt (1/t) and -dl/lam: 1/t is at hand, and in the predictor dl = -lam (1 + dt/t), so -dl/lam = 1 + dt/t --
        // no division per row (there were four IEEE divisions per row and iteration), one per lane after the reduction.
        double gmax = 0.0;
        double mu_aff = 0.0;
        double a_aff;
        if (active) {
        ROW_PASS_BEGIN();
        {
            const double dx = L.dv[mul24(kk, NV) + ZX], dy = L.dv[mul24(kk, NV) + ZY], dp = L.dv[mul24(kk, NV) + ZPSI];
            const double vx = L.v[mul24(kk, NV) + ZX], vy = L.v[mul24(kk, NV) + ZY], vp = L.v[mul24(kk, NV) + ZPSI];
            static_for<0, RPL>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                const bool a = ACT(s_);
                const double dt = row_dt(s_, dx, dy, dp, vx, vy, vp);
                if constexpr (!DIET_DT) dt_[DIET_DT ? 0 : s] = dt;
                const double q = dt * INVT(s);
                gmax = fmax(gmax, a ? fmax(-q, 1.0 + q) : 0.0);
            });
            gmax = blk_max<NTH>(gmax, L.scr, tl, 5);
            a_aff = gmax > 1.0 ? 1.0 / gmax : 1.0;          // min(1, alpha_max)
            ROW_PASS_BEGIN();
            static_for<0, RPL>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                if (ACT(s_)) {
                    const double dt = DIET_DT ? row_dt(s_, dx, dy, dp, vx, vy, vp) : dt_[DIET_DT ? 0 : s];
                    const double dl = -lam[s] - lam[s] * INVT(s) * dt;
                    mu_aff += (lam[s] + a_aff * dl) * (t[s] + a_aff * dt);
                }
            });
        }
        mu_aff = blk_sum<NTH>(mu_aff, L.scr, tl, 6) / m_rows;
        double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
        sigma = sigma * sigma * sigma;
        // ---- corrector rhs: gh = rg + sum c (qt + d rd) ----
        // Round 6: the predictor's right-hand side is still in gh (the factorisation and the solves only read it) and differs from the corrector's by
        // sum c (qt - lam) -- gh_pred = rg0 + sum c d rd, gh_corr = rg0 - sum lam c + sum c (qt + d rd) --, so the row pass ADDS that difference in place: no
        // copy of rg into gh, no barrier before the pass, no row residuals in it (-DTMPC_EXP_RHS_COPY rebuilds the copy form; the sums associate differently:
        // rounding level)
#ifdef TMPC_EXP_RHS_COPY
        for (int e = tl; e < (N + 1) * NV; e += NT) L.gh[e] = L.rg[e];
        team.sync();
#endif
        ROW_PASS_BEGIN();
        {
            double cs0 = 0, cs1 = 0, cs2 = 0;
            const double dx = L.dv[mul24(kk, NV) + ZX], dy = L.dv[mul24(kk, NV) + ZY], dp = L.dv[mul24(kk, NV) + ZPSI];   // predictor direction
            const double vx = L.v[mul24(kk, NV) + ZX], vy = L.v[mul24(kk, NV) + ZY], vp = L.v[mul24(kk, NV) + ZPSI];
            static_for<0, RPL>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                constexpr int K = C::template KIND<s>;
                const bool a = ACT(s_);
                const double dta = DIET_DT ? row_dt(s_, dx, dy, dp, vx, vy, vp) : dt_[DIET_DT ? 0 : s];
                const double dl = -lam[s] - lam[s] * INVT(s) * dta;
                qt[s] = a ? lam[s] + (dta * dl - sigma * mu) * INVT(s) : 0.0;
                double c0s, c1s, c2s;
                coef(s_, c0s, c1s, c2s);
#ifdef TMPC_EXP_RHS_COPY
                const double rr = rowdot(s_, c0s, c1s, c2s, vx, vy, vp, L.v + mul24(kk, NV)) - sb[s] - t[s];
                const double w = qt[s] + lam[s] * INVT(s) * (a ? rr : 0.0);
#else
                const double w = a ? qt[s] - lam[s] : 0.0;
#endif
                if constexpr (K != 1) { cs0 += w * c0s; cs1 += w * c1s; cs2 += w * c2s; }
                if constexpr (K != 0) { if (a && (K == 1 || (box >> s & 1))) lds_add(&L.gh[mul24(k, NV) + VARK(s_)], w * CUK(s_)); }
            });
            if constexpr (QSUM) {
                auto quad = [](double x) { x += dpp_move<0xB1, 0xf>(x, 0.0); x += dpp_move<0x4E, 0xf>(x, 0.0); if constexpr (LPS == 8) x += dpp_shift_zero<0x104>(x); return x; };
                cs0 = quad(cs0); cs1 = quad(cs1); cs2 = quad(cs2);
            }
            if constexpr (TSUM) {
                auto tri3 = [](double x) { const double t = x + dpp_shift_zero<0x130>(x); return x + dpp_shift_zero<0x130>(t); };
                cs0 = tri3(cs0); cs1 = tri3(cs1); cs2 = tri3(cs2);
            }
            if (stage_lane && (!QSUM || (c & (LPS == 8 ? 7 : 3)) == 0) && (!TSUM || c == 0 || c == 3)) { lds_add(&L.gh[mul24(k, NV) + ZX], cs0); lds_add(&L.gh[mul24(k, NV) + ZY], cs1); lds_add(&L.gh[mul24(k, NV) + ZPSI], cs2); }
        }
        team.sync();
        }
        pf.stop(PH_RHS);
        team.template solve<NTH, CP>(L, d, tl, 1 - sw, it, 2, active);
        pf.stop(PH_SOLVE);
        if (active) {
        gmax = 0.0;
        ROW_PASS_BEGIN();
        const double dxc = L.dv[mul24(kk, NV) + ZX], dyc = L.dv[mul24(kk, NV) + ZY], dpc = L.dv[mul24(kk, NV) + ZPSI];
        const double vxc = L.v[mul24(kk, NV) + ZX], vyc = L.v[mul24(kk, NV) + ZY], vpc = L.v[mul24(kk, NV) + ZPSI];
        static_for<0, RPL>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            const bool a = ACT(s_);
            const double dt = row_dt(s_, dxc, dyc, dpc, vxc, vyc, vpc);
            if constexpr (!DIET_DT) dt_[DIET_DT ? 0 : s] = dt;
            const double it_ = INVT(s);
            const double dl = -qt[s] - lam[s] * it_ * dt;
            gmax = fmax(gmax, a ? fmax(-dt * it_, -dl * rcp_nr(lam[s])) : 0.0);
        });
        gmax = blk_max<NTH>(gmax, L.scr, tl, 7);
        const double alpha = 0.999 > gmax ? 1.0 : 0.999 / gmax;        // min(1, 0.999 alpha_max)
        pf.stop(PH_ROWS);
        if (!isfinite(alpha)) { status = 4; active = false; }
        else if (alpha < 1e-12) { status = 3; active = false; }
        if (TEAM::NQ == 1 && !active) break;
        if (active) {
        ROW_PASS_BEGIN();
        static_for<0, RPL>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            if (ACT(s_)) {
                const double dt = DIET_DT ? row_dt(s_, dxc, dyc, dpc, vxc, vyc, vpc) : dt_[DIET_DT ? 0 : s];
                const double dl = -qt[s] - lam[s] * INVT(s) * dt;
                t[s] += alpha * dt; lam[s] += alpha * dl;
                if constexpr (!LEAN) invt_[s] = rcp_nr(t[s]);
            }
        });
        team.sync();                                 // the rows read v / dv above; v changes below
        for (int e = tl; e < (N + 1) * NV; e += NT) L.v[e] += alpha * L.dv[e];
        for (int e = tl; e < N * NX; e += NT) L.pq[NX + e] += alpha * L.dpi[NX + e];
        team.sync();
        }
        }
        pf.stop(PH_UPDATE);
    }
    *iters_out = iters;
    return status;
#undef stage_lane
#undef ROW_PASS_BEGIN
}

// Two-wave instantiations with 6 lanes per stage are built for two waves per SIMD (<= 256 registers): four trajectories
// per CU stay resident with two waves each.  The build refuses any instantiation that needs scratch.
template <int NLIN, int MM, int LPS, int NTH = 64, bool PROF = false, typename TEAM = Solo, int CM = 0>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu((NTH == 128 && LPS == 6 && NLIN == 8 && !PROF && std::is_same<TEAM, Solo>::value) ? 2 : 1,
                                                                    (NTH == 128 && LPS == 6 && NLIN == 8 && !PROF && std::is_same<TEAM, Solo>::value) ? 2 : 1)))
void tmpc_solve_fast_kernel(Dims d, int B, const double *__restrict__ xinit,
                                                             const double *__restrict__ x0, const double *__restrict__ params,
                                                             double *__restrict__ xtraj, double *__restrict__ utraj,
                                                             double *__restrict__ pobj, int *__restrict__ exit_code,
                                                             int *__restrict__ qp_status_out, int *__restrict__ sqp_iter_out,
                                                             double *__restrict__ res_eq_out, int *__restrict__ qp_iter_out,
                                                             long long *__restrict__ prof_out, StateIO io)
{
    using C = FastCfg<NLIN, MM, LPS>;
    constexpr int NT = NTH;
    const int NHk = C::RT ? d.n_up + d.M : C::NH, NLINk = C::RT ? d.n_up : NLIN;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= B) return;
    const int b = trajectory_of_block(blockIdx.x, B);
    if ((slot_flags(io, b) & ST_KEEP_ITERATE) && io.stopped[slot_of(io, b)]) return;      // this solver's loop has ended: outputs of its last call stand
    const Lds L = carve_fast(smem, d);
    const int N = d.N;
    const double *xi = xinit + (size_t)b * ext_nx(d);
    const double *pb_own = params + (size_t)b * N * d.npar;
    const double *pb = params + (size_t)param_base_of(io, b) * N * d.npar;
    // slack value: pinned by x_0 = xinit and slack' = 0 (tmpc_stage.hpp).  Re-read where it is used instead of being kept
    // in registers across the whole solve.
    auto slack_of = [&]() { return d.slack ? __builtin_nontemporal_load(xi + NX) : 0.0; };

    // loadWarmstart, or the iterate the handle holds; fresh or kept multipliers (StateIO, tmpc_solve.hip)
    for (int e = tid; e < (N + 1) * NV; e += NT) {
        const int k = e / NV, i = e - k * NV;
        L.z[e] = (slot_flags(io, b) & ST_KEEP_ITERATE) ? io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] : TMPC_LD_IN(x0 + ((size_t)b * (N + 1) + k) * ext_nv(d) + i);
    }
    for (int e = tid; e < (N + 1) * NX; e += NT) L.pi[e] = (slot_flags(io, b) & ST_KEEP_MULTIPLIERS) ? io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] : 0.0;
    for (int e = tid; e < N * NHk; e += NT) L.lamh[e] = (slot_flags(io, b) & ST_KEEP_MULTIPLIERS) ? io.lamh[(size_t)slot_of(io, b) * N * NHk + e] : 0.0;
    if (tid < 3) L.D[N * L.dstride + tid] = 0.0;    // zero triple read by box rows
    __syncthreads();
    if (tid < NU) L.z[N * NV + tid] = 0.0;
    __syncthreads();

    typename std::conditional<PROF, Prof, NoProf>::type pf;
    pf.init(prof_out);
    const long long t_begin = (PROF && prof_out) ? clock64() : 0;
    int status = 0, qp_status = 0, sqp_iter = 0, qp_iter_total = 0;
    double lam[C::RPL];
    for (int it = 0; it < d.n_sqp; it++) {
        pf.start();
        linearise<true, false, NTH, CM>(L, d, tid, pb, slack_of(), pb_own);
        __syncthreads();
        pf.stop(PH_LIN);
        int iters = 0;
        qp_status = ipm_fast<NLIN, MM, LPS, NTH, false>(L, d, tid, xi, &iters, pf, lam, TEAM());
        sqp_iter = it + 1; qp_iter_total += iters;
        if (qp_status != 0 && qp_status != 2) { status = 4; break; }
        status = 0;
        __syncthreads();
        int tid_w = tid;
        asm volatile("" : "+v"(tid_w));                   // opaque (see ipm_fast): keeps these addresses out of the prologue
        for (int e = tid_w; e < (N + 1) * NV; e += NT) {
            const int ks = e / NV, i = e - ks * NV;
            if (!(ks == N && i < NU)) L.z[e] += L.v[e];
        }
        for (int e = tid_w; e < N * NX; e += NT) L.pi[NX + e] = L.pq[NX + e];
        __syncthreads();
        // multipliers of the general rows for the next linearisation: (lam_upper - lam_lower) = -sgn lam
        constexpr int SPW = 64 / LPS;
        const int wl = tid_w & 63;
        const int k = (tid_w >> 6) * SPW + wl / LPS, c = wl % LPS;
        if (wl < SPW * LPS && k < N) {
#pragma unroll
            for (int s = 0; s < C::RPL; s++) {
                const int r = c + LPS * s;
                if (r < NHk) L.lamh[k * NHk + r] = (r < NLINk) ? lam[s] : -lam[s];
            }
        }
        __syncthreads();
        if (qp_status != 0) break;
    }
    if (io.flags & ST_STORE) {
        // (after a QP failure the staging area is stale; the host-side finalisation zeroes the multipliers of failed slots anyway,
        // as the reference resets a failed capsule, :187-191)
        for (int e = tid; e < (N + 1) * NV; e += NT) io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] = L.z[e];
        for (int e = tid; e < (N + 1) * NX; e += NT) io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] = L.pi[e];
        for (int e = tid; e < N * NHk; e += NT) io.lamh[(size_t)slot_of(io, b) * N * NHk + e] = L.lamh[e];
        if (tid == 0) { if (sqp_iter > 0) io.stopped[slot_of(io, b)] = qp_status != 0; io.valid[slot_of(io, b)] = 1; }
    }
    solve_epilogue<CM>(L, d, tid, b, xi, pb, slack_of(), status, qp_status, sqp_iter, qp_iter_total, xtraj, utraj, pobj, exit_code,
                       qp_status_out, sqp_iter_out, res_eq_out, qp_iter_out, prof_out, pf, t_begin, NTH);
}

// ---- compact kernel: eight trajectories per CU ---------------------------------------------------------------------------
// Same algorithm and arithmetic as tmpc_solve_fast_kernel (results are bitwise identical), laid out for two waves per SIMD:
// <= 256 registers (per-row values recomputed, lane indices re-made opaque per row pass) and <= 20 KB of LDS per trajectory
// (carve_compact).  Persistent workgroups: the launch has at most as many workgroups as the GPU holds resident, each takes
// trajectories from a ticket counter; the global NLP workspace is indexed by WORKGROUP, so it stays as small as the resident
// set (L2-resident) whatever the batch size.
// Work distribution of the persistent launch: ONE ticket counter, trajectories in batch order (adjacent trajectories -- a scene's
// guidance set -- run on different CUs and XCDs at the same time).  An XCD-aware variant was built and measured in round 3 (chunks of
// 64 consecutive trajectories dealt to the XCDs, one counter per XCD read through HW_REG_XCC_ID, exhausted XCDs stealing from the
// next): the L2 <-> fabric traffic did not move (FETCH 8.6 vs 9.0 GB, WRITE 2.87 GB per 32768-trajectory launch -- it is the
// per-workgroup workspace, not the parameter rows), a saturated launch ran as fast (833 vs 836 k solves/s) and a
// 4096-trajectory launch 4.5 % slower: rejected, profiles/round3_c_xcd_tickets_rejected.json.
template <int NLIN, int MM, int LPS, bool PROF = false, int NTH = 64, int CM = 0>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
void tmpc_solve_compact_kernel(Dims d, int B, const double *__restrict__ xinit,
                               const double *__restrict__ x0, const double *__restrict__ params,
                               double *__restrict__ xtraj, double *__restrict__ utraj,
                               double *__restrict__ pobj, int *__restrict__ exit_code,
                               int *__restrict__ qp_status_out, int *__restrict__ sqp_iter_out,
                               double *__restrict__ res_eq_out, int *__restrict__ qp_iter_out,
                               long long *__restrict__ prof_out, StateIO io)
{
    using C = FastCfg<NLIN, MM, LPS>;
    constexpr int NT = NTH;                         // 64: one wave per trajectory (N <= 21, three lanes per stage); 128: two waves (N <= 32, four lanes per stage)
    const int NHk = C::RT ? d.n_up + d.M : C::NH, NLINk = C::RT ? d.n_up : NLIN;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid0 = threadIdx.x;
    const int N = d.N;
    int tid = tid0;
    constexpr int CPV = compact_layout(NLIN, MM, NTH);           // layout parameter of ipm_fast and the Riccati routines (hoff<CP>)
    const Lds L = carve_compact(smem, io.ws + (size_t)blockIdx.x * ws_doubles(N, NTH > 64), d, NTH, CPV);
    ba_tab_init(L.tab, d, tid);
    if (tid < 3) L.D[N * L.dstride + tid] = 0.0;        // zero triple read by box rows (and as the third entry of packed rows)
    __syncthreads();
    for (;;) {
        int b = 0;
        if (tid == 0) b = atomicAdd(io.ticket, 1);
        if constexpr (NTH == 64) b = __builtin_amdgcn_readfirstlane(b);
        else {                                          // two waves: the ticket travels through LDS
            if (tid == 0) L.scr[0] = (double)b;
            __syncthreads();
            b = (int)L.scr[0];
            __syncthreads();                            // (scr is scratch of the solve below)
        }
        if (b >= B) break;
        asm volatile("" : "+v"(tid));                   // opaque per trajectory: per-lane addresses are recomputed, not kept live (and spilled) across solves
        if ((slot_flags(io, b) & ST_KEEP_ITERATE) && io.stopped[slot_of(io, b)]) continue;      // this solver's loop has ended: outputs of its last call stand
        const double *xi = xinit + (size_t)b * ext_nx(d);
        const double *pb_own = params + (size_t)b * N * d.npar;
        const double *pb = params + (size_t)param_base_of(io, b) * N * d.npar;
        auto slack_of = [&]() { return d.slack ? __builtin_nontemporal_load(xi + NX) : 0.0; };

        for (int e = tid; e < (N + 1) * NV; e += NT) {
            const int k = e / NV, i = e - k * NV;
            L.z[e] = (slot_flags(io, b) & ST_KEEP_ITERATE) ? io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] : TMPC_LD_IN(x0 + ((size_t)b * (N + 1) + k) * ext_nv(d) + i);
        }
        for (int e = tid; e < (N + 1) * NX; e += NT) L.pi[e] = (slot_flags(io, b) & ST_KEEP_MULTIPLIERS) ? io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] : 0.0;
        for (int e = tid; e < N * NHk; e += NT) L.lamh[e] = (slot_flags(io, b) & ST_KEEP_MULTIPLIERS) ? io.lamh[(size_t)slot_of(io, b) * N * NHk + e] : 0.0;
        __syncthreads();
        if (tid < NU) L.z[N * NV + tid] = 0.0;
        __syncthreads();

        typename std::conditional<PROF, Prof, NoProf>::type pf;
        pf.init(prof_out);
        const long long t_begin = (PROF && prof_out) ? clock64() : 0;
        int status = 0, qp_status = 0, sqp_iter = 0, qp_iter_total = 0;
        double lam[C::RPL];
        for (int it = 0; it < d.n_sqp; it++) {
            pf.start();
            TMPC_PRIO_LINEARISE();
            linearise<true, true, NTH, CM>(L, d, tid, pb, slack_of(), pb_own);
            __syncthreads();
            pf.stop(PH_LIN);
            int iters = 0;
            TMPC_PRIO_LOW();
            qp_status = ipm_fast<NLIN, MM, LPS, NTH, CPV>(L, d, tid, xi, &iters, pf, lam);
            sqp_iter = it + 1; qp_iter_total += iters;
            if (qp_status != 0 && qp_status != 2) { status = 4; break; }
            status = 0;
            __syncthreads();
            int tid_w = tid;
            asm volatile("" : "+v"(tid_w));
            for (int e = tid_w; e < (N + 1) * NV; e += NT) {
                const int ks = e / NV, i = e - ks * NV;
                if (!(ks == N && i < NU)) L.z[e] += L.v[e];
            }
            for (int e = tid_w; e < N * NX; e += NT) L.pi[NX + e] = L.pq[NX + e];
            __syncthreads();
            constexpr int SPW = 64 / LPS;
            const int wl = tid_w & 63;
            const int k = (tid_w >> 6) * SPW + wl / LPS, c = wl % LPS;
            if (wl < SPW * LPS && k < N) {
#pragma unroll
                for (int s = 0; s < C::RPL; s++) {
                    const int r = c + LPS * s;
                    if (r < NHk) L.lamh[k * NHk + r] = (r < NLINk) ? lam[s] : -lam[s];
                }
            }
            __syncthreads();
            if (qp_status != 0) break;
        }
        asm volatile("" : "+v"(tid));
        if (io.flags & ST_STORE) {
            for (int e = tid; e < (N + 1) * NV; e += NT) io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] = L.z[e];
            for (int e = tid; e < (N + 1) * NX; e += NT) io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] = L.pi[e];
            for (int e = tid; e < N * NHk; e += NT) io.lamh[(size_t)slot_of(io, b) * N * NHk + e] = L.lamh[e];
            if (tid == 0) { if (sqp_iter > 0) io.stopped[slot_of(io, b)] = qp_status != 0; io.valid[slot_of(io, b)] = 1; }
        }
        solve_epilogue<CM>(L, d, tid, b, xi, pb, slack_of(), status, qp_status, sqp_iter, qp_iter_total, xtraj, utraj, pobj, exit_code,
                       qp_status_out, sqp_iter_out, res_eq_out, qp_iter_out, prof_out, pf, t_begin, NTH);
        __syncthreads();                                // the next trajectory reuses LDS and the workspace
    }
}

}  // namespace tmpc
