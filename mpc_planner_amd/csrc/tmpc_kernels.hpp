#pragma once
// mpc_planner_amd/csrc/tmpc_kernels.hpp -- batched SQP_RTI solve kernels for gfx950 (MI355X): the kernel TEMPLATES of libtmpc_hip.so.
// Included by the translation units that instantiate them (tmpc_solve.hip: which instantiations live in which unit, tmpc_instances.hpp: the
// lists) and by tmpc_capi.hip (the C-ABI, which only takes their addresses).
//
// One workgroup (one 64-lane wavefront) owns one trajectory = one reference `Solver` instance
// (mpc_planner_modules/src/guidance_constraints.cpp:279-361 runs them as OpenMP threads; here they are
// workgroups of one launch).  All per-trajectory state of a solve -- iterate, multipliers, the stage blocks
// [W g | B A b | D beta], the interior-point rows and the Riccati factors -- lives in LDS for the whole
// solve; HBM is touched only for the inputs (xinit, warm start, parameter rows) and the outputs.
//
// Phases per RTI iteration (Solver::solve, acados_solver_interface.cpp:86-119):
//   1. linearise     lane k = stage k: dynamics + sensitivities, cost/rows + derivatives, Lagrangian Hessian,
//                    MIRROR (registers), stage block -> LDS
//   2. QP            Mehrotra predictor-corrector IPM; per iteration: residuals, barrier Hessian, square-root
//                    Riccati factorisation (backward sweep over stages, lanes over matrix entries, wave
//                    shuffles inside the 7x7 Cholesky), two Riccati vector solves, row updates, wave reductions
//   3. full step     z += dz, multipliers from the QP
// then completeOneIteration (:162-204): cost, trajectories, res_eq, exit-code mapping.
//
// Parts: tmpc_stage.hpp (stage functions: dynamics, cost, rows, MIRROR), tmpc_riccati.hpp (Riccati factorisation + vector sweeps),
// tmpc_fast.hpp (the register-resident "fast" and compact solve kernels), tmpc_scan.hpp (parallel-in-time Newton solve of the latency
// variants); this file holds the LDS layout, wave helpers, the linearisation and the generic solve kernel.  The C-ABI (kernel dispatch
// tables, the handle and every exported entry point of include/tmpc_hip.h) is its own translation unit, tmpc_capi.hip, together with the
// small kernels of tmpc_aux_kernels.hpp (selection, records, f-1 / f-2 / f-3 helpers).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/tmpc_hip.h"
#include "tmpc_stage.hpp"

namespace tmpc {

constexpr int NT = 64;   // threads per trajectory (one wavefront)

static __constant__ int c_pi[NP28] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6};
static __constant__ int c_pj[NP28] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6};

// ---- persistent solver state (tmpc_solve_iterations) ---------------------------------------------------------------
// The reference's acados capsules keep the NLP iterate and its multipliers between calls: solveOneIteration continues from
// them and loadWarmstart overwrites the primal part only (acados_solver_interface.cpp:67-77,121-160,274-284; SURVEY Appendix
// D-4).  Here that state lives in HBM per trajectory slot; a solve launch optionally starts from it and writes it back.
enum { ST_KEEP_ITERATE = 1, ST_KEEP_MULTIPLIERS = 2, ST_STORE = 4, ST_COMPLETE = 8 };
struct StateIO {
    double *z;          // [B][(N+1) NV]  iterate
    double *pi;         // [B][(N+1) NX]  dynamics multipliers
    double *lamh;       // [B][N nh]      (lam_upper - lam_lower) of the general rows, kernel row order [topology | slack | ellipsoids]
    int *stopped;       // [B]            1: this slot's RTI loop has ended (a QP stopped with qp_status != 0, :105-106)
    int flags;
    // compact kernels only (tmpc_fast.hpp): per-workgroup NLP workspace in global memory (L2-resident: one slot per RESIDENT
    // workgroup, not per trajectory) and the work ticket of the persistent launch
    double *ws;         // [grid][ws_doubles(N)]
    int *ticket;        // next trajectory to solve (zeroed before the launch)
    const int *slot;    // [B] state slot of every batch entry (tmpc_set_slots); nullptr: entry b uses slot b
    int *valid;         // [B_max] the slot holds state of an earlier call; the others start fresh whatever the flags say
    const int *share;   // [B] tmpc_set_param_sharing: entry b reads every parameter but its own halfspace rows from entry share[b]; nullptr: none
};
// Parameter rows of batch entry b: the row block everything but the topology / scenario halfspaces is read from.  A guidance set's
// planners carry copies of the main solver's parameters (guidance_constraints.cpp:300 `*solver = *_solver`) and differ in their own
// halfspaces only: reading the copies from ONE of them keeps the set's parameter footprint in L2 at 1/64 (results are bitwise the same).
__device__ __forceinline__ int param_base_of(const StateIO &io, int b) { return io.share ? io.share[b] : b; }
// State slot of batch entry b: by default b itself; callers that keep one slot per Solver and launch a changing subset of them
// (GuidanceConstraints with a varying number of guidance trajectories) give the map with tmpc_set_slots.
__device__ __forceinline__ int slot_of(const StateIO &io, int b) { return io.slot ? io.slot[b] : b; }
// The keep-flags apply to slots that have state: a slot that was never stored starts like a fresh capsule.
__device__ __forceinline__ int slot_flags(const StateIO &io, int b)
{
    if (!(io.flags & (ST_KEEP_ITERATE | ST_KEEP_MULTIPLIERS))) return io.flags;
    return io.valid[slot_of(io, b)] ? io.flags : (io.flags & ~(ST_KEEP_ITERATE | ST_KEEP_MULTIPLIERS));
}

// ---- per-trajectory LDS layout (doubles) ----------------------------------------------------
struct Lds {
    double *z, *pi, *W, *g, *BA, *b, *D, *beta;          // NLP iterate + stage blocks of the current QP
    double *t, *lam, *invt, *qt;                         // interior-point rows
    double *v, *pq, *Hh, *rg, *gh, *rb, *dv, *dpi, *pr, *y, *rdiag, *scr;
    double *dyn8;                                        // 8 non-constant entries of [B A] per stage
    double *lamh;                                        // fast kernel: staged (lam_upper - lam_lower) of the general rows
    int nh, NG, GB, XB, nrows;
    // compact layout (tmpc_fast.hpp, carve_compact): z, pi, W, g, b point into the GLOBAL workspace; [B A] is not stored -- `tab` holds the 8 non-constant entries per stage
    // followed by 16 constants (ba_tab below); the rows' Jacobians are packed (pairs for topology rows, triples otherwise)
    double *tab;
    int n_pair, dstride;
    double *scan;                                        // fast layout, latency mode 2: scratch of the parallel-in-time solve (tmpc_scan.hpp), behind the layout                                 // rows r < n_pair store (gx, gy) only; doubles per stage in D
};

// ---- sparse [B A] (compact kernels) ------------------------------------------------------------------------------
// For the unicycle [B A] (5 x 7) has 8 stage-dependent entries (dyn8, tmpc_riccati.hpp) and constants 0, 1, dt, dt^2/2.  `tab` =
// dyn8[N][8] followed by 20 constants; an entry is addressed by a 4-bit code: 0..7 = dyn8 entry of the stage, 8..13 = 0, 1, dt,
// dt^2/2 and the spline row's own (sdt, shdt2) -- (dt, dt^2/2) for the contouring model, (0, 0) for SecondOrderUnicycleModel, whose fifth
// state slot is inert (Dims::model).  The last 12 constants are rows psi, v, s of [B A] in dyn8 column order (a, w, psi, v) for the forward sweep.
// Reading [B A] through the table returns exactly the values the dense copy held (zeros and ones included), so every sum that
// runs over a row or column of [B A] keeps its operation order: results are bitwise those of the dense layout.
constexpr int BAC_0 = 8, BAC_1 = 9, BAC_DT = 10, BAC_H = 11, BAC_SDT = 12, BAC_SH = 13, BA_NGROUP0 = 8, BA_NCONST = 20;
constexpr unsigned ba_pack(int a, int w, int x, int y, int p, int v, int s_)
{
    return (unsigned)a | (unsigned)w << 4 | (unsigned)x << 8 | (unsigned)y << 12 | (unsigned)p << 16 | (unsigned)v << 20 | (unsigned)s_ << 24;
}
// row m of [B A]: codes of its 7 columns (a, w, x, y, psi, v, s)
__device__ __forceinline__ constexpr unsigned ba_rowcode(int m)
{
    return m == 0 ? ba_pack(0, 1, BAC_1, BAC_0, 2, 3, BAC_0)
         : m == 1 ? ba_pack(4, 5, BAC_0, BAC_1, 6, 7, BAC_0)
         : m == 2 ? ba_pack(BAC_0, BAC_DT, BAC_0, BAC_0, BAC_1, BAC_0, BAC_0)
         : m == 3 ? ba_pack(BAC_DT, BAC_0, BAC_0, BAC_0, BAC_0, BAC_1, BAC_0)
                  : ba_pack(BAC_SH, BAC_0, BAC_0, BAC_0, BAC_0, BAC_SDT, BAC_1);
}
// offset (doubles) of entry (m, j) of stage k in `tab`
__device__ __forceinline__ int ba_off(int N, int k, int m, int j)
{
    const unsigned rc = m == 0 ? ba_rowcode(0) : m == 1 ? ba_rowcode(1) : m == 2 ? ba_rowcode(2) : m == 3 ? ba_rowcode(3) : ba_rowcode(4);
    const int code = (int)((rc >> (4 * j)) & 15u);
    return code < 8 ? k * 8 + code : N * 8 + code - 8;
}
__device__ __forceinline__ void ba_tab_init(double *tab, const Dims &d, int tid)
{
    if (tid < BA_NCONST) {
        const double dt = d.dt, h = d.hdt2, sdt = d.sdt, sh = d.shdt2;
        //                          0    1    dt  h  sdt  sh  (pad)     | psi: a  w   psi  v  | v: a   w    psi  v  | s: a   w    psi  v
        const double c[BA_NCONST] = {0.0, 1.0, dt, h, sdt, sh, 0.0, 0.0,   0.0, dt, 1.0, 0.0,   dt, 0.0, 0.0, 1.0,   sh, 0.0, 0.0, sdt};
        double val = 0.0;
#pragma unroll
        for (int i = 0; i < BA_NCONST; i++) if (i == tid) val = c[i];
        tab[d.N * 8 + tid] = val;
    }
}
// doubles of one workgroup's global workspace
__host__ __device__ inline int ws_doubles(int N, bool two_wave = false)      // z, pi, W, g, b (+ the two-wave kernels' second share of W)
{
    return (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + (N + 1) * NV + (N + 1) * NX + (two_wave ? N * NP28 : 0);
}


__host__ __device__ inline int lds_doubles(int N, int nh)
{
    const int nrows = N * nh + 4 * N + 10 * (N - 1);
    int n = 0;
    n += (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + (N + 1) * NV + N * NX * NV + N * NX + N * nh * 3 + N * nh;
    n += 4 * nrows;
    n += (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + 2 * (N + 1) * NV + N * NX + (N + 1) * NV + (N + 1) * NX +
         (N + 1) * NX + N * NU + N * NU + 64 + N * 8;
    return n;
}

// Dims::split_rows.  The one-wave kernels' linearisation exchanges the helpers' share of the rows' Hessian (2 N x 6 doubles) and MIRROR's 3 x 3 blocks
// (N x 8 doubles) through L.dv, which every layout follows with L.dpi: 12 (N + 1) contiguous doubles >= 12 N.  The staging region (beta, lamh: 2 N nh
// doubles from the start of the work region) is being written at that time and must end before L.dv.  Offset of L.dv from the start of the work region:
// (N + 1) (NV + NX + hstride + NV + NV) + N NX with hstride >= NP28 (carve_fast, carve_compact) -- the bare stride gives the tightest bound, so one
// answer holds for the fast and all compact layouts of a shape.
__host__ __device__ inline bool split_rows_for(int N, int nh)
{
    return 2 * N * nh <= (N + 1) * (NV + NX + NP28 + NV + NV) + N * NX && 12 * (N + 1) >= 12 * N;
}

__device__ __forceinline__ Lds carve(double *s, const Dims &d)
{
    Lds L;
    const int N = d.N;
    L.nh = d.n_up + d.M;
    L.NG = N * L.nh; L.GB = L.NG; L.XB = L.NG + 4 * N; L.nrows = L.XB + 10 * (N - 1);
    auto take = [&](int n) { double *p = s; s += n; return p; };
    L.z = take((N + 1) * NV); L.pi = take((N + 1) * NX); L.W = take((N + 1) * NP28); L.g = take((N + 1) * NV);
    L.BA = take(N * NX * NV); L.b = take(N * NX); L.D = take(N * L.nh * 3); L.beta = take(N * L.nh);
    L.t = take(L.nrows); L.lam = take(L.nrows); L.invt = take(L.nrows); L.qt = take(L.nrows);
    L.v = take((N + 1) * NV); L.pq = take((N + 1) * NX); L.Hh = take((N + 1) * NP28);
    L.rg = take((N + 1) * NV); L.gh = take((N + 1) * NV); L.rb = take(N * NX); L.dv = take((N + 1) * NV);
    L.dpi = take((N + 1) * NX); L.pr = take((N + 1) * NX); L.y = take(N * NU); L.rdiag = take(N * NU);
    L.scr = take(64); L.dyn8 = take(N * 8);
    L.lamh = nullptr;
    return L;
}

// ---- workgroup -> trajectory ------------------------------------------------------------------------
// The dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2), and trajectories of one scene
// (adjacent in the batch) share 92 % of their parameter rows.  Giving every XCD a contiguous range of trajectories was
// measured (round 1): HBM fetch per launch 127 -> 99 MB, but kernel time 10.5 -> 11.3 ms -- a scene's trajectories need
// similar iteration counts, so whole slow scenes pile up on one XCD while others drain.  The kernel is compute-bound
// (0.15 % of HBM peak), so the identity mapping, which interleaves every scene over all XCDs, stays.
__device__ __forceinline__ int trajectory_of_block(int blk, int B)
{
    (void)B;
    return blk;
}

// ---- wave reductions (one wavefront per workgroup) -----------------------------------------------
// DPP row shifts / row broadcasts on the two 32-bit halves (pure VALU, no LDS round trips as with ds_bpermute): after
// row_shr 1,2,4,8 lane 15 of every 16-lane row holds the row's result, row_bcast:15 folds rows 0->1 and 2->3, row_bcast:31
// folds the lower half into the upper one; lane 63 then holds the wave's result and is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double x, double identity)
{
    union { double d; int i[2]; } u, o;
    u.d = x; o.d = identity;
    o.i[0] = __builtin_amdgcn_update_dpp(o.i[0], u.i[0], CTRL, ROW_MASK, 0xf, false);
    o.i[1] = __builtin_amdgcn_update_dpp(o.i[1], u.i[1], CTRL, ROW_MASK, 0xf, false);
    return o.d;
}
// row shift whose vacated lanes read 0 (bound_ctrl:0): for sums no identity value has to be materialised first
template <int CTRL>
__device__ __forceinline__ double dpp_shift_zero(double x)
{
    union { double d; int i[2]; } u, o;
    u.d = x;
    o.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], CTRL, 0xf, 0xf, true);
    o.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], CTRL, 0xf, 0xf, true);
    return o.d;
}
template <typename Op>
__device__ __forceinline__ double wave_reduce(double x, double identity, Op op)
{
    x = op(x, dpp_move<0x111, 0xf>(x, identity));      // row_shr:1
    x = op(x, dpp_move<0x112, 0xf>(x, identity));      // row_shr:2
    x = op(x, dpp_move<0x114, 0xf>(x, identity));      // row_shr:4
    x = op(x, dpp_move<0x118, 0xf>(x, identity));      // row_shr:8
    x = op(x, dpp_move<0x142, 0xa>(x, identity));      // row_bcast:15 into rows 1 and 3
    x = op(x, dpp_move<0x143, 0xc>(x, identity));      // row_bcast:31 into rows 2 and 3
    union { double d; int i[2]; } u;
    u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], 63);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], 63);
    return u.d;
}
__device__ __forceinline__ double wave_max(double x)
{
    return wave_reduce(x, -__builtin_huge_val(), [](double a, double b) { return fmax(a, b); });
}
__device__ __forceinline__ double wave_min(double x)
{
    return wave_reduce(x, __builtin_huge_val(), [](double a, double b) { return fmin(a, b); });
}
__device__ __forceinline__ double wave_sum(double x)
{
    return wave_reduce(x, 0.0, [](double a, double b) { return a + b; });
}

// Workgroup reductions for the multi-wave variants of the fast kernel (NTH = 128: two waves, 256: four): wave reduction, then one LDS exchange.
// `scr` = L.scr (64 doubles); slots [slot * NW, slot * NW + NW) are used (NW = NTH / 64).  NTH == 64 reduces to the wave reduction.
template <int NTH, typename Op>
__device__ __forceinline__ double blk_combine(double x, double *scr, int tid, int slot, Op op)
{
    if constexpr (NTH == 64) return x;
    constexpr int NW = NTH / 64;
    // every call site has its own slot and is reached once per interior-point iteration, with barriers in between:
    // the previous readers of the slot are long done
    if ((tid & 63) == 0) scr[slot * NW + (tid >> 6)] = x;
    __syncthreads();
    double r = scr[slot * NW];
#pragma unroll
    for (int w = 1; w < NW; w++) r = op(r, scr[slot * NW + w]);      // (fixed order: wave 0, 1, ...)
    return r;
}
template <int NTH> __device__ __forceinline__ double blk_max(double x, double *scr, int tid, int slot = 0)
{
    return blk_combine<NTH>(wave_max(x), scr, tid, slot, [](double a, double b) { return fmax(a, b); });
}
template <int NTH> __device__ __forceinline__ double blk_min(double x, double *scr, int tid, int slot = 0)
{
    return blk_combine<NTH>(wave_min(x), scr, tid, slot, [](double a, double b) { return fmin(a, b); });
}
template <int NTH> __device__ __forceinline__ double blk_sum(double x, double *scr, int tid, int slot = 0)
{
    return blk_combine<NTH>(wave_sum(x), scr, tid, slot, [](double a, double b) { return a + b; });
}

// The convergence quantities of an interior-point iteration at once.  The four residual norms (stationarity, dynamics, rows, complementarity) are
// used for exactly two decisions -- "all finite?" and "all <= qp_tol?" -- and both are decisions about their MAXIMUM, so each lane folds its four
// partial maxima into one before the wave reduction (round 5: one wave_max instead of four, ~60 VALU instructions per interior-point iteration;
// the decisions, and with them every result, are unchanged).  `worst` returns that maximum, `mu` the complementarity sum.  Multi-wave kernels: one
// LDS exchange and one barrier (slots 0 .. 2 NW - 1 of scr: the blk_* call sites use slots >= 5).
template <int NTH>
__device__ __forceinline__ void blk_residuals(double &worst, double g, double b, double dd, double m, double &mu, double *scr, int tid)
{
    worst = wave_max(fmax(fmax(g, b), fmax(dd, m))); mu = wave_sum(mu);
    if constexpr (NTH > 64) {
        constexpr int NW = NTH / 64;
        if ((tid & 63) == 0) { const int w = tid >> 6; scr[w] = worst; scr[NW + w] = mu; }
        __syncthreads();
        double v[2 * NW];
#pragma unroll
        for (int i = 0; i < 2 * NW; i++) v[i] = scr[i];
        worst = v[0]; mu = v[NW];
#pragma unroll
        for (int w = 1; w < NW; w++) { worst = fmax(worst, v[w]); mu = mu + v[NW + w]; }
    }
}

// ---- interior-point row access --------------------------------------------------------------------
struct Row { int k, var, general; double sgn; };   // general: index into D/beta; box: var = z index

__device__ __forceinline__ Row row_decode(const Lds &L, const Dims &d, int r)
{
    Row R;
    if (r < L.NG) {
        R.k = r / L.nh; const int j = r - R.k * L.nh;
        R.general = r; R.var = -1; R.sgn = (j < d.n_up) ? -1.0 : 1.0;    // topology / slack rows: upper 0; ellipsoids: lower 1
    } else if (r < L.XB) {
        const int q = r - L.GB;
        R.k = q >> 2; R.var = (q >> 1) & 1; R.general = -1; R.sgn = (q & 1) ? -1.0 : 1.0;
    } else {
        const int q = r - L.XB;
        R.k = 1 + q / 10; const int rem = q - (R.k - 1) * 10;
        R.var = 2 + (rem >> 1); R.general = -1; R.sgn = (rem & 1) ? -1.0 : 1.0;
    }
    return R;
}
__device__ __forceinline__ double row_dot(const Lds &L, const Row &R, const double *vec)
{
    const double *vk = vec + R.k * NV;
    if (R.general >= 0) {
        const double *Dr = L.D + R.general * 3;
        return Dr[0] * vk[ZX] + Dr[1] * vk[ZY] + Dr[2] * vk[ZPSI];
    }
    return vk[R.var];
}
__device__ __forceinline__ double row_beta(const Lds &L, const Dims &d, const Row &R)
{
    if (R.general >= 0) return L.beta[R.general];
    return (R.sgn > 0.0 ? d.lb[R.var] : d.ub[R.var]) - L.z[R.k * NV + R.var];
}

// ---- cross-lane helpers ---------------------------------------------------------------------------
// Broadcast lane `src` (wave-uniform) of a double through two v_readlane_b32: no LDS, no bpermute.
__device__ __forceinline__ double readlane_d(double x, int src)
{
    union { double d; int i[2]; } u;
    u.d = x;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
}
// Broadcast lane LANE of every 16-lane row to the whole row: ONE v_mov_b64_dpp row_newbcast (the only DPP control the f64 ALU
// supports).  Unlike v_readlane the value stays in a VGPR: no SGPR-pair operand limit on its consumers, no VALU -> SGPR -> VALU
// hazard waits, and the four rows of a wave stay independent.  The Riccati sweeps use rows of 16 lanes: lanes 0..6 of row 0
// hold the trajectory, the other rows compute on copies and are ignored.
template <int LANE>
__device__ __forceinline__ double bcast16(double x)
{
    static_assert(LANE >= 0 && LANE < 16, "row_newbcast lane");
    const long long r = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long long, x), 0x150 + LANE, 0xf, 0xf, false);   // (no `old` value to materialise)
    return __builtin_bit_cast(double, r);
}
// compile-time loop: f(std::integral_constant<int, I>) for I = A .. B-1 (lane numbers of DPP controls must be immediates)
template <int A, int B, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (A < B) { f(std::integral_constant<int, A>{}); static_for<A + 1, B>(f); }
}
// Index arithmetic of the sequential sweeps.  A 32-bit integer multiply (v_mul_lo_u32, and the v_mad_u64_u32 the compiler picks for
// `uniform * constant + per-lane offset`) is a QUARTER-rate instruction: 16 cycles of a SIMD that issues an f64 FMA in 4.  The sweeps' stage loops
// had 2-5 of them per stage step (5 of the factor loop's 138 instructions, but 80 of its ~630 issue cycles).  mul24: per-lane stride x stage
// index, both far below 2^24 -> v_mul_u32_u24 / v_mad_u32_u24 (full rate).  uni: a wave-uniform product is kept in an SGPR (s_mul_i32) and only
// added on the vector side.
__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
#ifdef TMPC_EXP_NO_UNI
__device__ __forceinline__ int uni(int x) { return x; }
#else
__device__ __forceinline__ int uni(int x) { asm volatile("" : "+s"(x)); return x; }
#endif
// 1/sqrt(d) for d > 0: v_rsq_f64 seed (5e-8 relative, measured) + one third-order (Halley) step: with e = 1 - d y^2,
// y (1 + e/2 + 3 e^2/8) leaves an error of order e^3 -- full double precision in five dependent operations, where two Newton steps
// take eight (this sits on the critical chain of the Cholesky: seven pivots per stage)
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

}  // namespace tmpc
#include "tmpc_riccati.hpp"
namespace tmpc {

// gh = rg + sum_rows sgn c (qt + d rd),  d = lam/t, rd = sgn (c.v - beta) - t ; predictor: qt = lam
__device__ void build_rhs(const Lds &L, const Dims &d, int tid, bool predictor)
{
    const int N = d.N;
    for (int it = tid; it < (N + 1) * NV; it += NT) {
        const int k = it / NV, i = it - k * NV;
        double acc = L.rg[it];
        if (i < NU) {
            if (k < N) {
                for (int side = 0; side < 2; side++) {
                    const int r = L.GB + k * 4 + i * 2 + side;
                    const double sgn = side ? -1.0 : 1.0;
                    const double beta = (side ? d.ub[i] : d.lb[i]) - L.z[k * NV + i];
                    const double rd = sgn * (L.v[k * NV + i] - beta) - L.t[r];
                    const double q = predictor ? L.lam[r] : L.qt[r];
                    acc += sgn * (q + L.lam[r] * L.invt[r] * rd);
                }
            }
        } else if (k >= 1 && k < N) {
            for (int side = 0; side < 2; side++) {
                const int r = L.XB + (k - 1) * 10 + (i - NU) * 2 + side;
                const double sgn = side ? -1.0 : 1.0;
                const double beta = (side ? d.ub[i] : d.lb[i]) - L.z[k * NV + i];
                const double rd = sgn * (L.v[k * NV + i] - beta) - L.t[r];
                const double q = predictor ? L.lam[r] : L.qt[r];
                acc += sgn * (q + L.lam[r] * L.invt[r] * rd);
            }
        }
        if (k < N && i >= ZX && i <= ZPSI) {
            const double *vk = L.v + k * NV;
            for (int j = 0; j < L.nh; j++) {
                const int r = k * L.nh + j;
                const double sgn = (j < d.n_up) ? -1.0 : 1.0;
                const double *Dr = L.D + r * 3;
                const double cv = Dr[0] * vk[ZX] + Dr[1] * vk[ZY] + Dr[2] * vk[ZPSI];
                const double rd = sgn * (cv - L.beta[r]) - L.t[r];
                const double q = predictor ? L.lam[r] : L.qt[r];
                acc += sgn * Dr[i - ZX] * (q + L.lam[r] * L.invt[r] * rd);
            }
        }
        L.gh[it] = acc;
    }
    __syncthreads();
}

// ---- optional in-kernel phase profile (debug entry point tmpc_debug_profile) --------------------------
enum { PH_LIN = 0, PH_RES, PH_HH, PH_FACTOR, PH_RHS, PH_SOLVE, PH_ROWS, PH_UPDATE, PH_FINAL, PH_TOTAL, PH_COUNT };
struct Prof {
    long long *out; long long acc[PH_COUNT]; long long t0;
    __device__ __forceinline__ void init(long long *o) { out = o; for (int i = 0; i < PH_COUNT; i++) acc[i] = 0; }
    __device__ __forceinline__ void start() { if (out) t0 = clock64(); }
    __device__ __forceinline__ void stop(int ph) { if (out) { const long long t1 = clock64(); acc[ph] += t1 - t0; t0 = t1; } }
    __device__ __forceinline__ void finish(int tid, int b, long long t_begin)
    {
        if (!out) return;
        stop(PH_FINAL);
        acc[PH_TOTAL] = clock64() - t_begin;
        if (tid == 0) for (int i = 0; i < PH_COUNT; i++) out[(size_t)b * PH_COUNT + i] = acc[i];
    }
};
// Production instantiations of the fast kernel carry no profiling state (the 10 phase accumulators cost ~20 registers).
struct NoProf {
    __device__ __forceinline__ void init(long long *) {}
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void stop(int) {}
    __device__ __forceinline__ void finish(int, int, long long) {}
};

// One QP solve.  Returns status (0 ok, 2 max iter, 3 min step, 4 NaN); *iters = IPM iterations.
__device__ int ipm_solve(const Lds &L, const Dims &d, int tid, int *iters_out, Prof &pf)
{
    const int N = d.N;
    const double m_rows = (double)L.nrows;
    // cold start: v = 0 (dx_0 = xinit - x_0 is already in v[0]), pi = 0, t = max(r, thr0), lam = mu0/t
    for (int r = tid; r < L.nrows; r += NT) {
        const Row R = row_decode(L, d, r);
        const double rr = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R));
        const double t = rr > d.thr0 ? rr : d.thr0;
        L.t[r] = t; L.invt[r] = 1.0 / t; L.lam[r] = d.mu0 / t;
    }
    __syncthreads();
    int status = 2, iters = 0;
    for (int it = 0;; it++) {
        // ---------------- residuals ----------------
        pf.start();
        double res_g = 0.0, res_b = 0.0, res_d = 0.0, res_m = 0.0, mu = 0.0;
        for (int e = tid; e < (N + 1) * NV; e += NT) {
            const int k = e / NV, i = e - k * NV;
            double acc = 0.0;
            const bool skip = (k == N && i < NU) || (k == 0 && i >= NU);
            if (!skip) {
                acc = L.g[e];
                const double *Wk = L.W + k * NP28; const double *vk = L.v + k * NV;
#pragma unroll
                for (int j = 0; j < NV; j++) acc += Wk[sidx(i, j)] * vk[j];
                if (k < N) {
                    const double *BA = L.BA + k * NX * NV;
#pragma unroll
                    for (int l = 0; l < NX; l++) acc += BA[l * NV + i] * L.pq[(k + 1) * NX + l];
                }
                if (i >= NU && k >= 1) acc -= L.pq[k * NX + i - NU];
                // - sum sgn lam c_i
                if (i < NU) {
                    const int r = L.GB + k * 4 + i * 2;
                    acc += -L.lam[r] + L.lam[r + 1];
                } else if (k < N) {      // k >= 1 here
                    const int r = L.XB + (k - 1) * 10 + (i - NU) * 2;
                    acc += -L.lam[r] + L.lam[r + 1];
                }
                if (k < N && i >= ZX && i <= ZPSI)
                    for (int j = 0; j < L.nh; j++) {
                        const int r = k * L.nh + j;
                        const double sgn = (j < d.n_up) ? -1.0 : 1.0;
                        acc -= sgn * L.lam[r] * L.D[r * 3 + i - ZX];
                    }
            }
            L.rg[e] = acc;
            res_g = fmax(res_g, fabs(acc));
        }
        for (int e = tid; e < N * NX; e += NT) {
            const int k = e / NX, i = e - k * NX;
            double acc = L.b[e] - L.v[(k + 1) * NV + NU + i];
            const double *BA = L.BA + k * NX * NV + i * NV; const double *vk = L.v + k * NV;
#pragma unroll
            for (int j = 0; j < NV; j++) acc += BA[j] * vk[j];
            L.rb[e] = acc;
            res_b = fmax(res_b, fabs(acc));
        }
        for (int r = tid; r < L.nrows; r += NT) {
            const Row R = row_decode(L, d, r);
            const double rd = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R)) - L.t[r];
            const double comp = L.lam[r] * L.t[r];
            res_d = fmax(res_d, fabs(rd)); res_m = fmax(res_m, comp); mu += comp;
        }
        res_g = wave_max(res_g); res_b = wave_max(res_b); res_d = wave_max(res_d); res_m = wave_max(res_m);
        mu = wave_sum(mu) / m_rows;
        __syncthreads();
        pf.stop(PH_RES);
        if (!(isfinite(res_g) && isfinite(res_b) && isfinite(res_d) && isfinite(res_m))) { status = 4; break; }
        if (res_g <= d.qp_tol && res_b <= d.qp_tol && res_d <= d.qp_tol && res_m <= d.qp_tol) { status = 0; break; }
        if (it >= d.qp_iter_max) { status = 2; break; }
        iters = it + 1;

        // ---------------- barrier-augmented Hessian ----------------
        for (int e = tid; e < (N + 1) * NP28; e += NT) {
            const int k = e / NP28, pe = e - k * NP28;
            const int i = c_pi[pe], j = c_pj[pe];
            double acc = L.W[e];
            if (i == j) {
                if (i < NU) {
                    if (k < N) { const int r = L.GB + k * 4 + i * 2; acc += L.lam[r] * L.invt[r] + L.lam[r + 1] * L.invt[r + 1]; }
                } else if (k >= 1 && k < N) {
                    const int r = L.XB + (k - 1) * 10 + (i - NU) * 2;
                    acc += L.lam[r] * L.invt[r] + L.lam[r + 1] * L.invt[r + 1];
                }
            }
            if (k < N && j >= ZX && i <= ZPSI)       // i >= j: both in {x, y, psi}
                for (int q = 0; q < L.nh; q++) {
                    const int r = k * L.nh + q;
                    acc += L.lam[r] * L.invt[r] * L.D[r * 3 + i - ZX] * L.D[r * 3 + j - ZX];
                }
            L.Hh[e] = acc;
        }
        __syncthreads();
        pf.stop(PH_HH);
        const bool fbad = riccati_factor<NT>(L, d, tid);
        pf.stop(PH_FACTOR);
        if (fbad) { status = 4; break; }

        // ---------------- predictor ----------------
        build_rhs(L, d, tid, true);
        pf.stop(PH_RHS);
        riccati_solve<NT>(L, d, tid);
        pf.stop(PH_SOLVE);
        double amax = 1e300;
        for (int r = tid; r < L.nrows; r += NT) {
            const Row R = row_decode(L, d, r);
            const double rd = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R)) - L.t[r];
            const double dt = R.sgn * row_dot(L, R, L.dv) + rd;
            const double dl = -L.lam[r] - L.lam[r] * L.invt[r] * dt;
            if (dt < 0.0) amax = fmin(amax, -L.t[r] / dt);
            if (dl < 0.0) amax = fmin(amax, -L.lam[r] / dl);
            L.qt[r] = dt * dl;                      // keep dt_aff * dlam_aff for the corrector
        }
        double a_aff = fmin(1.0, wave_min(amax));
        double mu_aff = 0.0;
        for (int r = tid; r < L.nrows; r += NT) {
            const Row R = row_decode(L, d, r);
            const double rd = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R)) - L.t[r];
            const double dt = R.sgn * row_dot(L, R, L.dv) + rd;
            const double dl = -L.lam[r] - L.lam[r] * L.invt[r] * dt;
            mu_aff += (L.lam[r] + a_aff * dl) * (L.t[r] + a_aff * dt);
        }
        mu_aff = wave_sum(mu_aff) / m_rows;
        double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
        sigma = sigma * sigma * sigma;
        // ---------------- corrector ----------------
        for (int r = tid; r < L.nrows; r += NT)
            L.qt[r] = L.lam[r] + (L.qt[r] - sigma * mu) * L.invt[r];          // q / t
        __syncthreads();
        pf.stop(PH_ROWS);
        build_rhs(L, d, tid, false);
        pf.stop(PH_RHS);
        riccati_solve<NT>(L, d, tid);
        pf.stop(PH_SOLVE);
        amax = 1e300;
        for (int r = tid; r < L.nrows; r += NT) {
            const Row R = row_decode(L, d, r);
            const double rd = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R)) - L.t[r];
            const double dt = R.sgn * row_dot(L, R, L.dv) + rd;
            const double dl = -L.qt[r] - L.lam[r] * L.invt[r] * dt;
            if (dt < 0.0) amax = fmin(amax, -L.t[r] / dt);
            if (dl < 0.0) amax = fmin(amax, -L.lam[r] / dl);
        }
        const double alpha = fmin(1.0, 0.999 * wave_min(amax));
        pf.stop(PH_ROWS);
        if (!isfinite(alpha)) { status = 4; break; }
        if (alpha < 1e-12) { status = 3; break; }
        // ---------------- update ----------------
        for (int r = tid; r < L.nrows; r += NT) {
            const Row R = row_decode(L, d, r);
            const double rd = R.sgn * (row_dot(L, R, L.v) - row_beta(L, d, R)) - L.t[r];
            const double dt = R.sgn * row_dot(L, R, L.dv) + rd;
            const double dl = -L.qt[r] - L.lam[r] * L.invt[r] * dt;
            const double tn = L.t[r] + alpha * dt;
            L.t[r] = tn; L.invt[r] = 1.0 / tn; L.lam[r] += alpha * dl;
        }
        __syncthreads();     // rows read v/dv above; v changes below
        for (int e = tid; e < (N + 1) * NV; e += NT) L.v[e] += alpha * L.dv[e];
        for (int e = tid; e < N * NX; e += NT) L.pq[NX + e] += alpha * L.dpi[NX + e];
        __syncthreads();
        pf.stop(PH_UPDATE);
    }
    *iters_out = iters;
    return status;
}

// four-wave linearisation: parts of stage_linearise on wave 1 | waves 2, 3 (-DTMPC_EXP_QUAD_ROWS_ON_COST_WAVE: the first form, halfspace / scenario rows next to the cost)
#ifdef TMPC_EXP_QUAD_ROWS_ON_COST_WAVE
#define TMPC_QUAD_COST_PART 4
#define TMPC_QUAD_ROWS_PART 5
#else
#define TMPC_QUAD_COST_PART 6
#define TMPC_QUAD_ROWS_PART 7
#endif
// the latency kernels' (two / four waves per trajectory) 4 x 4 block: paired round-robin sweep (mirror_n, tmpc_stage.hpp)
#ifdef TMPC_EXP_NO_PAIR4
constexpr bool MIRROR_PAIR = false;
#else
constexpr bool MIRROR_PAIR = true;
#endif
// MIRROR of the one-wave kernels (round 5).  Lane k < N linearises stage k; the other lanes of the wave are idle copies.  With a zero disc
// offset the Lagrangian Hessian is block diagonal under {a, w, psi, v} | {x, y, spline} (mirror7), and the two blocks' Jacobi iterations are
// independent: lane k keeps the 4 x 4 block, lane N + k takes the 3 x 3 block of stage k -- padded to 4 x 4 with a zero row / column, which the
// cyclic sweep skips (a_pq = 0) and which changes neither the sweep's convergence sums nor the reconstruction (+ 0.0) -- so ONE mirror_n<4> call
// regularises both blocks of all stages at once, where mirror7 ran mirror_n<4> and then mirror_n<3> on 20 of 64 lanes.  Bitwise what mirror7
// computes.  A stage whose W couples the blocks (any cross entry != 0: the curvature-aware cost, a disc offset) takes the 7 x 7 iteration on its own
// lane as before.  `xch`: N * 8 doubles of LDS that nothing else uses during the linearisation (the caller checks), exchanged under two barriers.
__device__ __forceinline__ void mirror7_pair(double (*A)[NV], double eps, int lane, int N, bool owner, double *xch)
{
    constexpr int IA[4] = {ZA, ZW, ZPSI, ZV}, IB[3] = {ZX, ZY, ZS};
    bool coupled = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) coupled |= (A[IA[i]][IB[j]] != 0.0) | (A[IB[j]][IA[i]] != 0.0);
    if (owner) {
        double *x = xch + lane * 8;
        x[0] = A[ZX][ZX]; x[1] = A[ZY][ZX]; x[2] = A[ZY][ZY]; x[3] = A[ZS][ZX]; x[4] = A[ZS][ZY]; x[5] = A[ZS][ZS];
        x[6] = coupled ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool partner = lane >= N && lane < 2 * N;
    double M[4][4];
    bool skip = coupled;                                   // (idle lanes carry stage N - 1's W like its owner: they follow that lane)
    if (partner) {
        const double *x = xch + (lane - N) * 8;
        M[0][0] = x[0]; M[1][0] = M[0][1] = x[1]; M[1][1] = x[2]; M[2][0] = M[0][2] = x[3]; M[2][1] = M[1][2] = x[4]; M[2][2] = x[5];
#pragma unroll
        for (int i = 0; i < 4; i++) { M[i][3] = 0.0; M[3][i] = 0.0; }
        skip = x[6] != 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) M[i][j] = A[IA[i]][IA[j]];
    }
    if (!skip) mirror_n<4>(M, eps);
    if (partner && !skip) {
        double *x = xch + (lane - N) * 8;
        x[0] = M[0][0]; x[1] = M[1][0]; x[2] = M[1][1]; x[3] = M[2][0]; x[4] = M[2][1]; x[5] = M[2][2];
    }
    __syncthreads();
    if (!coupled) {
        if (!partner) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) A[IA[i]][IA[j]] = M[i][j];
        }
        if (owner) {
            const double *x = xch + lane * 8;
            A[ZX][ZX] = x[0]; A[ZY][ZX] = A[ZX][ZY] = x[1]; A[ZY][ZY] = x[2]; A[ZS][ZX] = A[ZX][ZS] = x[3]; A[ZS][ZY] = A[ZY][ZS] = x[4]; A[ZS][ZS] = x[5];
        }
    } else {
        double F[NV][NV];
#pragma unroll
        for (int i = 0; i < NV; i++)
#pragma unroll
            for (int j = 0; j < NV; j++) F[i][j] = A[i][j];
        mirror_n<NV>(F, eps);
#pragma unroll
        for (int i = 0; i < NV; i++)
#pragma unroll
            for (int j = 0; j < NV; j++) A[i][j] = F[i][j];
    }
}

// ---- stage linearisation by lane k --------------------------------------------------------------
// NTH = 128 (fast layout, two waves per trajectory; hand-written stages): the stage evaluation is split over the waves -- wave 0 the dynamics
// (rollout with sensitivities, [B A], the multipliers' share of the Hessian) and half of the ellipsoid rows, wave 1 the cost, the halfspace
// rows and the other ellipsoids -- which run as different code at the same time; the shares of W are exchanged through LDS and the
// regularisation (MIRROR: more than half of a stage's chain) is shared too.  W = W_0 + W_1 associates differently from the one-wave sum (rounding level).
template <bool FAST, bool CP = false, int NTH = 64, int CM = 0>
__device__ __forceinline__ void linearise(const Lds &L, const Dims &d, int tid, const double *params, double slack, const double *params_own = nullptr)
{
    const int N = d.N;
#ifndef TMPC_GENERATED_STAGE
    if constexpr (FAST && NTH == 256) {
        // Four waves per trajectory (round 6, the control-tick kernel: a whole CU serves one trajectory, one wave per SIMD).  The stage evaluation is
        // split FOUR ways, by what is computed rather than by stage: wave 0 the dynamics (rollout with sensitivities, [B A], defects, the multipliers'
        // share of the Hessian), wave 1 the cost, waves 2 and 3 the rows (halfspace, scenario / decomp, obstacle) -- three lanes per stage each, so a
        // lane evaluates at most ceil(n / 6) rows of each class.  The shares of W meet in LDS (the scan scratch behind the layout is dead while the stage blocks are
        // built); MIRROR's two diagonal blocks then run on waves 0 and 1 as in the two-wave kernels.  W = W_dyn + W_cost + sum of the six obstacle
        // shares, in that order (associates differently from the one-wave sum: rounding level).  Fast layout only.  N <= 21: three lanes per stage on the
        // obstacle waves (six shares); 22 <= N <= 32 (the shipped jackal / jackalsimulator horizon, N = 30): two (four shares).
        static_assert(!CP, "four-wave linearisation: fast layout");
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int wv = tid_l >> 6, ln = tid_l & 63;
        const int G = 3 * N <= 64 ? 3 : 2;                         // lane groups of a wave
        const int grp = ln >= 2 * N ? 2 : (ln >= N ? 1 : 0);       // lane group inside the wave: 0 owner, 1 / 2 helpers (waves 2, 3 only)
        const bool owner = ln < N;                                  // (full EXEC mask: other lanes redo a stage and do not store)
        const bool inrng = ln < G * N;
        const bool rowlane = wv >= 2 && inrng;                      // an obstacle-row lane
        const int k = inrng ? ln - grp * N : N - 1;
        double z[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) z[i] = L.z[k * NV + i];
        const double *p = params + (size_t)k * d.npar;
        const long long own_delta = params_own ? (long long)(params_own - params) : 0;
        const int nh = L.nh;
        double W[NV][NV], g[NV], BA[NX * NV], xn[NX];
        auto lamh = [&](int r) { return L.lamh[k * nh + r]; };
        const bool writes = wv >= 2 ? rowlane : owner;
        auto sink = [&](int r, const RowOut &ro) {
            if (writes) {
                const double sg = (r < d.n_up) ? -1.0 : 1.0;     // fast layouts keep the SIGNED row Jacobian (ipm_fast reads it as it is)
                double *Dr = L.D + k * L.dstride + 3 * r;
                Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy; Dr[2] = sg * ro.gp;
                const double bound = (r < d.n_up || cm_gaussian_rows(CM)) ? 0.0 : 1.0;
                L.beta[k * nh + r] = bound - ro.h;
            }
        };
        // exchange region: [0, N 28): wave 0's share of W (as in the two-wave kernels); [N 28, N 28 + 2 G N 6): the obstacle lanes' shares (x, y, psi block)
        double *W0s = L.scan + k * NP28;
        double *Wes = L.scan + N * NP28;
        if (wv == 0) {                                           // dynamics
            stage_linearise<CM>(d, z, p, 1, L.pi[(k + 1) * NX + 0], L.pi[(k + 1) * NX + 1], lamh, sink, W, g, BA, xn, slack, nullptr, own_delta, 3);
            if (owner) {
#pragma unroll
                for (int i = 0; i < NX * NV; i++) L.BA[k * NX * NV + i] = BA[i];
                double *d8 = L.dyn8 + k * 8;
                d8[D8_XA] = BA[0 * NV + ZA]; d8[D8_XW] = BA[0 * NV + ZW]; d8[D8_XP] = BA[0 * NV + ZPSI]; d8[D8_XV] = BA[0 * NV + ZV];
                d8[D8_YA] = BA[1 * NV + ZA]; d8[D8_YW] = BA[1 * NV + ZW]; d8[D8_YP] = BA[1 * NV + ZPSI]; d8[D8_YV] = BA[1 * NV + ZV];
#pragma unroll
                for (int i = 0; i < NX; i++) L.b[k * NX + i] = xn[i] - L.z[(k + 1) * NV + NU + i];
#pragma unroll
                for (int i = 0; i < NV; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) W0s[pidx(i, j)] = W[i][j];
            }
        } else if (wv == 1) {                                    // the cost
            stage_linearise<CM>(d, z, p, 1, 0.0, 0.0, lamh, sink, W, g, BA, xn, slack, nullptr, own_delta, TMPC_QUAD_COST_PART);
            if (owner) {
#pragma unroll
                for (int i = 0; i < NV; i++) L.g[k * NV + i] = g[i];
#pragma unroll
                for (int i = 0; i < NV; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) L.W[k * NP28 + pidx(i, j)] = W[i][j];
            }
        } else {                                                 // the rows (halfspace, scenario / decomp, obstacle): lane (wave, group) takes rows first, first + 2 G, ... of each class
            const int first_ = (wv - 2) * G + grp;
            stage_linearise<CM>(d, z, p, 1, 0.0, 0.0, lamh, sink, W, g, BA, xn, slack, nullptr, own_delta, TMPC_QUAD_ROWS_PART, [&]() { return first_; }, 2 * G, true);
            if (rowlane) {
                double *x = Wes + (first_ * N + k) * 6;
                x[0] = W[ZX][ZX]; x[1] = W[ZX][ZY]; x[2] = W[ZY][ZY]; x[3] = W[ZX][ZPSI]; x[4] = W[ZY][ZPSI]; x[5] = W[ZPSI][ZPSI];
            }
        }
        __syncthreads();                                         // every share of W is in LDS
        if (wv >= 2) { __syncthreads(); return; }               // (the obstacle waves are done: they only attend the barrier below)
        {
            double w0[NP28], w1[NP28];
#pragma unroll
            for (int e = 0; e < NP28; e++) { w0[e] = W0s[e]; w1[e] = L.W[k * NP28 + e]; }
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) { W[i][j] = w0[pidx(i, j)] + w1[pidx(i, j)]; W[j][i] = W[i][j]; }
            double e6[6];
#pragma unroll
            for (int q = 0; q < 6; q++) e6[q] = 0.0;
#pragma unroll
            for (int s6 = 0; s6 < 6; s6++) {
                if (s6 >= 2 * G) break;
                const double *x = Wes + (s6 * N + k) * 6;
#pragma unroll
                for (int q = 0; q < 6; q++) e6[q] += x[q];
            }
            W[ZX][ZX] += e6[0]; W[ZY][ZY] += e6[2]; W[ZPSI][ZPSI] += e6[5];
            W[ZX][ZY] += e6[1]; W[ZY][ZX] += e6[1];
            W[ZX][ZPSI] += e6[3]; W[ZPSI][ZX] += e6[3];
            W[ZY][ZPSI] += e6[4]; W[ZPSI][ZY] += e6[4];
        }
        __syncthreads();                                         // ... and read by both waves: the W slot may be overwritten
        constexpr int IA[4] = {ZA, ZW, ZPSI, ZV}, IB[3] = {ZX, ZY, ZS};
        bool coupled = false;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) coupled |= (W[IA[i]][IB[j]] != 0.0) | (W[IB[j]][IA[i]] != 0.0);
        if (wv == 0) {
            if (coupled) {
                mirror_n<NV>(W, d.reg_eps);
                if (owner) {
#pragma unroll
                    for (int i = 0; i < NV; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) L.W[k * NP28 + pidx(i, j)] = W[i][j];
                }
            } else {
                double Ba[4][4];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) Ba[i][j] = W[IA[i]][IA[j]];
                mirror_n<4, MIRROR_PAIR>(Ba, d.reg_eps);
                if (owner) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) if (IA[i] >= IA[j]) L.W[k * NP28 + pidx(IA[i], IA[j])] = Ba[i][j];
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) L.W[k * NP28 + sidx(IA[i], IB[j])] = 0.0;      // (the cross entries: exactly zero here)
                }
            }
            if (tid_l == N) {                                // terminal node: zero cost, no rows: MIRROR(0) = eps I on the state block
                const int wN = N * NP28, gN = N * NV;
                for (int e = 0; e < NP28; e++) L.W[wN + e] = 0.0;
                for (int i = NU; i < NV; i++) L.W[wN + pidx(i, i)] = d.reg_eps;
                for (int i = 0; i < NV; i++) L.g[gN + i] = 0.0;
            }
        } else if (!coupled) {
            double Bb[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Bb[i][j] = W[IB[i]][IB[j]];
            mirror_n<3>(Bb, d.reg_eps);
            if (owner) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) if (IB[i] >= IB[j]) L.W[k * NP28 + pidx(IB[i], IB[j])] = Bb[i][j];
            }
        }
        return;
    }
    if constexpr (FAST && NTH == 128) {                  // (both layouts: the compact one differs in where the blocks go, not in the arithmetic)
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int wv = tid_l >> 6, ln = tid_l & 63;
        const bool owner = ln < N;
        const int k = owner ? ln : N - 1;                   // (full EXEC mask: lanes >= N redo stage N - 1 and do not store)
        double z[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) z[i] = L.z[k * NV + i];
        const double *p = params + (size_t)k * d.npar;
        const long long own_delta = params_own ? (long long)(params_own - params) : 0;
        const int nh = L.nh;
        double W[NV][NV], g[NV], BA[NX * NV], xn[NX];
        auto lamh = [&](int r) { return L.lamh[k * nh + r]; };
        auto sink = [&](int r, const RowOut &ro) {
            if (owner) {
                const double sg = (r < d.n_up) ? -1.0 : 1.0;     // fast layouts keep the SIGNED row Jacobian (ipm_fast reads it as it is)
                if constexpr (CP) {                               // packed: (gx, gy) for topology rows, triples for the others
                    double *Dr = L.D + k * L.dstride + (r < L.n_pair ? 2 * r : 3 * r - L.n_pair);
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy;
                    if (r >= L.n_pair) Dr[2] = sg * ro.gp;
                } else {
                    double *Dr = FAST ? L.D + k * L.dstride + 3 * r : L.D + (k * nh + r) * 3;      // (fast layout: padded stage stride, carve_fast)
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy; Dr[2] = sg * ro.gp;
                }
                const double bound = (r < d.n_up || cm_gaussian_rows(CM)) ? 0.0 : 1.0;     // ellipsoid rows: h >= 1; Gaussian rows: h >= 0
                L.beta[k * nh + r] = bound - ro.h;
            }
        };
        // both waves park their share of W (wave 1 in the stage's W slot, wave 0 in the -- idle -- residual arrays of the interior-point
        // work region), so that after the barrier each of them has the complete W and MIRROR can be shared as well: with a zero disc offset
        // W is block diagonal under {a, w, psi, v} | {x, y, spline} (mirror7), and the two blocks are regularised on different waves --
        // bitwise what mirror7 computes.  A coupled W (any cross entry != 0) takes the 7 x 7 iteration on wave 0.
        double *W0s = L.scan + k * NP28;                     // (N * NP28 doubles behind the layout: every two-wave launch allocates them; compact layout: in the global workspace)
        if (wv == 1) {                                       // cost, halfspace rows, second half of the ellipsoid rows
            stage_linearise<CM>(d, z, p, 1, 0.0, 0.0, lamh, sink, W, g, BA, xn, slack, nullptr, own_delta, 2);
            if (owner) {
#pragma unroll
                for (int i = 0; i < NV; i++) L.g[k * NV + i] = g[i];
#pragma unroll
                for (int i = 0; i < NV; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) L.W[k * NP28 + pidx(i, j)] = W[i][j];
            }
        } else {                                             // dynamics, first half of the ellipsoid rows
            stage_linearise<CM>(d, z, p, 1, L.pi[(k + 1) * NX + 0], L.pi[(k + 1) * NX + 1], lamh, sink, W, g, BA, xn, slack, nullptr, own_delta, 1);
            if (owner) {
                if constexpr (!CP) {
#pragma unroll
                    for (int i = 0; i < NX * NV; i++) L.BA[k * NX * NV + i] = BA[i];
                }
                double *d8 = (CP ? L.tab : L.dyn8) + k * 8;
                d8[D8_XA] = BA[0 * NV + ZA]; d8[D8_XW] = BA[0 * NV + ZW]; d8[D8_XP] = BA[0 * NV + ZPSI]; d8[D8_XV] = BA[0 * NV + ZV];
                d8[D8_YA] = BA[1 * NV + ZA]; d8[D8_YW] = BA[1 * NV + ZW]; d8[D8_YP] = BA[1 * NV + ZPSI]; d8[D8_YV] = BA[1 * NV + ZV];
#pragma unroll
                for (int i = 0; i < NX; i++) L.b[k * NX + i] = xn[i] - L.z[(k + 1) * NV + NU + i];
#pragma unroll
                for (int i = 0; i < NV; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) W0s[pidx(i, j)] = W[i][j];
            }
        }
        __syncthreads();                                     // both shares of W are in LDS
        {
            double w0[NP28], w1[NP28];
#pragma unroll
            for (int e = 0; e < NP28; e++) { w0[e] = W0s[e]; w1[e] = L.W[k * NP28 + e]; }
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) { W[i][j] = w0[pidx(i, j)] + w1[pidx(i, j)]; W[j][i] = W[i][j]; }
        }
        __syncthreads();                                     // ... and read by both waves: the W slot may be overwritten
        constexpr int IA[4] = {ZA, ZW, ZPSI, ZV}, IB[3] = {ZX, ZY, ZS};
        bool coupled = false;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) coupled |= (W[IA[i]][IB[j]] != 0.0) | (W[IB[j]][IA[i]] != 0.0);
        if (wv == 0) {
            if (coupled) {
                mirror_n<NV>(W, d.reg_eps);
                if (owner) {
#pragma unroll
                    for (int i = 0; i < NV; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) L.W[k * NP28 + pidx(i, j)] = W[i][j];
                }
            } else {
                double Ba[4][4];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) Ba[i][j] = W[IA[i]][IA[j]];
                mirror_n<4, MIRROR_PAIR>(Ba, d.reg_eps);
                if (owner) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) if (IA[i] >= IA[j]) L.W[k * NP28 + pidx(IA[i], IA[j])] = Ba[i][j];
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) L.W[k * NP28 + sidx(IA[i], IB[j])] = 0.0;      // (the cross entries: exactly zero here)
                }
            }
            if (tid_l == N) {                                // terminal node: zero cost, no rows: MIRROR(0) = eps I on the state block
                const int wN = N * NP28, gN = N * NV;
                for (int e = 0; e < NP28; e++) L.W[wN + e] = 0.0;
                for (int i = NU; i < NV; i++) L.W[wN + pidx(i, i)] = d.reg_eps;
                for (int i = 0; i < NV; i++) L.g[gN + i] = 0.0;
            }
        } else if (!coupled) {
            double Bb[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) Bb[i][j] = W[IB[i]][IB[j]];
            mirror_n<3>(Bb, d.reg_eps);
            if (owner) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) if (IB[i] >= IB[j]) L.W[k * NP28 + pidx(IB[i], IB[j])] = Bb[i][j];
            }
        }
        return;
    }
#endif
    // Every lane runs the (register-hungry) stage evaluation with the full EXEC mask -- lanes >= N redo stage N-1 and
    // simply do not store -- so that no spill/reload of live registers happens under a partial mask.
    int tid_l = tid;
    asm volatile("" : "+v"(tid_l));                       // opaque: no hoisting of per-stage addresses out of the RTI loop
    const bool owner = tid_l < N;
    // Round 5, one-wave fast / compact kernels: lanes N .. 3N-1 do not idle on a copy of stage N-1 any more -- lane g N + k (g = 1, 2) linearises a copy
    // of stage k as well and the three lanes of a stage take every third ROW of each class (obstacle rows: ~75 instructions and seven L2-latency
    // parameter loads per row, a fifth of the linearisation's time at 8 rows; halfspace and scenario rows); the helpers write their rows' Jacobians themselves and
    // hand their share of the rows' Hessian to the owner through LDS (their W is that share and nothing else: everything else that enters W is
    // multiplied by zero on a helper).  The rest of the stage (dynamics, cost, halfspace rows) is computed redundantly by all three -- same
    // instructions, no extra issue slots.  The sum of the three shares associates differently from the sequential
    // sum over the rows (rounding level).  `split`: the wave has the lanes and the exchange buffer lies clear of the staging region -- a fact of the
    // SHAPE, computed once on the host (split_rows_for below) so that the fast and the compact kernel of a shape always take the same path.
#ifdef TMPC_GENERATED_STAGE
    const bool split = false;                            // (emitted stage functions evaluate all their rows in one piece: tmpc_gen::rows has no notion of a share)
#else
    const bool split = FAST && NTH == 64 && 3 * N <= 64 && L.nh >= 3 && d.split_rows;       // (Dims::split_rows: the exchange buffer -- dv and dpi, 12 (N + 1) contiguous doubles -- lies clear of the staging region in EVERY layout of the shape)
#endif
    const bool helper = split && tid_l >= N && tid_l < 3 * N;
    const int k = owner ? tid_l : (helper ? (tid_l >= 2 * N ? tid_l - 2 * N : tid_l - N) : N - 1);
    auto ell_first = [&]() { return split ? (tid_l >= N ? 1 : 0) + (tid_l >= 2 * N ? 1 : 0) : 0; };     // the lane's group: owner 0, helpers 1 and 2
    {
        double z[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) z[i] = L.z[k * NV + i];
        const double *p = params + (size_t)k * d.npar;     // Solver_acados_update_params(k, all_parameters[k*NP])
        const long long own_delta = params_own ? (long long)(params_own - params) : 0;   // (shared rows: where the trajectory's own halfspaces are)
        double W[NV][NV], g[NV], BA[NX * NV], xn[NX];
        const int nh = L.nh;
        auto lamh = [&](int r) {                            // (lam_upper - lam_lower) of the previous QP
            if (FAST) return L.lamh[k * nh + r];
            const double sgn = (r < d.n_up) ? -1.0 : 1.0;
            return -sgn * L.lam[k * nh + r];
        };
        auto sink = [&](int r, const RowOut &ro) {
            if (owner || helper) {                            // (every row is evaluated by exactly one of a stage's three lanes, which writes it)
                // fast layouts (the register-row kernels) keep the SIGNED row Jacobian sgn D -- upper-bounded rows -1, lower-bounded +1 -- so that
                // the row passes of ipm_fast read their coefficients as they are; the generic kernel keeps D and applies the sign itself
                const double sg = FAST ? ((r < d.n_up) ? -1.0 : 1.0) : 1.0;
                if constexpr (CP) {
                    // packed Jacobians: (gx, gy) for topology rows (gp == 0 exactly, lin_row_eval), triples for the others
                    double *Dr = L.D + k * L.dstride + (r < L.n_pair ? 2 * r : 3 * r - L.n_pair);
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy;
                    if (r >= L.n_pair) Dr[2] = sg * ro.gp;
                } else {
                    double *Dr = FAST ? L.D + k * L.dstride + 3 * r : L.D + (k * nh + r) * 3;      // (fast layout: padded stage stride, carve_fast)
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy; Dr[2] = sg * ro.gp;
                }
                const double bound = (r < d.n_up || cm_gaussian_rows(CM)) ? 0.0 : 1.0;     // ellipsoid rows: h >= 1; Gaussian rows: h >= 0
                L.beta[k * nh + r] = bound - ro.h;
            }
        };
        stage_linearise<CM>(d, z, p, 1, L.pi[(k + 1) * NX + 0], L.pi[(k + 1) * NX + 1], lamh, sink, W, g, BA, xn, slack,
                        L.W + k * NP28, own_delta, 0, ell_first, split ? 3 : 1, helper);   // (generated solvers park the cost Hessian in the stage's W slot)
        if (split) {                                        // the helpers' W = their share of the obstacle rows' Hessian (x, y, psi block) -> the owner
            double *xe = L.dv;                              // 2 N x 6 doubles (dv and dpi are contiguous: 252 doubles at N = 20)
            if (helper) {
                double *x = xe + (tid_l - N) * 6;                      // ((group - 1) N + k)
                x[0] = W[ZX][ZX]; x[1] = W[ZX][ZY]; x[2] = W[ZY][ZY]; x[3] = W[ZX][ZPSI]; x[4] = W[ZY][ZPSI]; x[5] = W[ZPSI][ZPSI];
            }
            __syncthreads();
            if (owner) {
                const double *x1 = xe + k * 6, *x2 = xe + (N + k) * 6;
                const double e0 = x1[0] + x2[0], e1 = x1[1] + x2[1], e2 = x1[2] + x2[2], e3 = x1[3] + x2[3], e4 = x1[4] + x2[4], e5 = x1[5] + x2[5];
                W[ZX][ZX] += e0; W[ZY][ZY] += e2; W[ZPSI][ZPSI] += e5;
                W[ZX][ZY] += e1; W[ZY][ZX] += e1;
                W[ZX][ZPSI] += e3; W[ZPSI][ZX] += e3;
                W[ZY][ZPSI] += e4; W[ZPSI][ZY] += e4;
            }
            __syncthreads();                                // (the exchange buffer is MIRROR's next)
        }
        // everything but W leaves the registers BEFORE the register-hungry MIRROR
        // compact layout: g, b, W live in the global workspace (same [stage][entry] layout: a lane's stores of one array share
        // one address register and differ in the immediate offset); [B A] is kept as its 8 non-constant entries only
        constexpr int es = 1;
        const int gk = k * NV, bk = k * NX, wk = k * NP28;
        if (owner) {
#pragma unroll
            for (int i = 0; i < NV; i++) L.g[gk + i * es] = g[i];
            if constexpr (!CP) {
#pragma unroll
                for (int i = 0; i < NX * NV; i++) L.BA[k * NX * NV + i] = BA[i];
            }
            double *d8 = (CP ? L.tab : L.dyn8) + k * 8;
            d8[D8_XA] = BA[0 * NV + ZA]; d8[D8_XW] = BA[0 * NV + ZW]; d8[D8_XP] = BA[0 * NV + ZPSI]; d8[D8_XV] = BA[0 * NV + ZV];
            d8[D8_YA] = BA[1 * NV + ZA]; d8[D8_YW] = BA[1 * NV + ZW]; d8[D8_YP] = BA[1 * NV + ZPSI]; d8[D8_YV] = BA[1 * NV + ZV];
#pragma unroll
            for (int i = 0; i < NX; i++) L.b[bk + i * es] = xn[i] - L.z[(k + 1) * NV + NU + i];
        }
        // MIRROR: the two diagonal blocks of every stage in different lanes at the same time where the wave has the lanes (2 N <= 64) and the
        // exchange buffer -- the tail of the interior-point work region, dead while the stage blocks are built -- lies clear of the staging
        // region (beta, lamh) the rows are being written into; the generic kernel keeps the one-lane form
        if (FAST && NTH == 64 && 2 * N <= 64 && d.split_rows) mirror7_pair(W, d.reg_eps, tid_l, N, owner, L.dv);
        else mirror7(W, d.reg_eps);
        if (owner) {
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) L.W[wk + pidx(i, j) * es] = W[i][j];
        }
    }
    if (tid == N) {
        // terminal node: zero cost, no rows: MIRROR(0) = eps I on the state block
        constexpr int es = 1;
        const int wN = N * NP28, gN = N * NV;
        for (int e = 0; e < NP28; e++) L.W[wN + e * es] = 0.0;
        for (int i = NU; i < NV; i++) L.W[wN + pidx(i, i) * es] = d.reg_eps;
        for (int i = 0; i < NV; i++) L.g[gN + i * es] = 0.0;
    }
}

// ---- completeOneIteration (acados_solver_interface.cpp:162-204): cost, trajectories, res_eq, exit-code mapping ----
template <int CM = 0, typename PF>
__device__ __forceinline__ void solve_epilogue(const Lds &L, const Dims &d, int tid, int b, const double *xi, const double *pb, double slack, int status,
                               int qp_status, int sqp_iter, int qp_iter_total, double *xtraj, double *utraj, double *pobj,
                               int *exit_code, int *qp_status_out, int *sqp_iter_out, double *res_eq_out, int *qp_iter_out,
                               long long *prof_out, PF &pf, long long t_begin, int nth = NT)
{
    const int N = d.N;
    pf.start();
    double cost = 0.0, res = 0.0;
    {   // full EXEC (lanes >= N redo stage N-1 and discard)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int ks = tid_e < N ? tid_e : N - 1;
        double z[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) z[i] = L.z[ks * NV + i];
        double cval;
#ifndef TMPC_GENERATED_STAGE
        if constexpr (cm_curvature_aware(CM)) { CostOutCA co; cost_eval_ca(d, z, pb + (size_t)ks * d.npar, 1, co, false, slack); cval = co.val; }
        else
#endif
        { CostOut co; cost_eval(d, z, pb + (size_t)ks * d.npar, 1, co, false, slack); cval = co.val; }
        DynOut dy;
        dyn_eval(d, z, dy, false);
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < NX; i++) r = fmax(r, fabs(dy.xn[i] - L.z[(ks + 1) * NV + NU + i]));
        if (tid < N) { cost = d.dt * cval; res = r; }
    }
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    if (tid_o < NX) res = fmax(res, fabs(L.z[NU + tid_o] - xi[tid_o]));
    cost = wave_sum(cost); res = wave_max(res);              // contributions live in lanes < N + NX <= 64: wave 0 holds the totals
    const int nxe = ext_nx(d);
    for (int e = tid_o; e < (N + 1) * nxe; e += nth) {
        const int k = d.slack ? e / (NX + 1) : e / NX, i = e - k * nxe;     // (two divisions by constants: a division by the run-time nxe keeps its
                                                                             //  reciprocal live across the persistent kernels' whole trajectory loop)
        TMPC_ST_OUT(xtraj + (size_t)b * (N + 1) * nxe + e, i < NX ? L.z[k * NV + NU + i] : slack);      // the pinned slack state
    }
    for (int e = tid_o; e < N * NU; e += nth) {
        const int k = e / NU, i = e - k * NU;
        TMPC_ST_OUT(utraj + (size_t)b * N * NU + e, L.z[k * NV + i]);
    }
    if (tid == 0) {
        if (res > 1e-2 && status == 0) status = 4;
        if (!isfinite(cost)) status = 4;
        pobj[b] = cost; res_eq_out[b] = res;
        exit_code[b] = status == 0 ? 1 : (status == 1 ? 0 : status);      // Forces-style mapping (:197-201)
        if (d.n_sqp > 0) { qp_status_out[b] = qp_status; sqp_iter_out[b] = sqp_iter; qp_iter_out[b] = qp_iter_total; }   // (an evaluation-only call keeps the statistics of the iterations before it)
    }
    (void)prof_out;
    pf.finish(tid, b, t_begin);
}

// ---- the solve kernel ---------------------------------------------------------------------------
template <int CM>      // stage model (stage_model(): Dims::cost_model + 2 * Dims::row_model): 0 MPCC contouring + ellipsoids, 1 curvature-aware contouring, 2 Gaussian rows
__global__ __launch_bounds__(NT) void tmpc_solve_kernel(Dims d, int B, const double *__restrict__ xinit,
                                                        const double *__restrict__ x0, const double *__restrict__ params,
                                                        double *__restrict__ xtraj, double *__restrict__ utraj,
                                                        double *__restrict__ pobj, int *__restrict__ exit_code,
                                                        int *__restrict__ qp_status_out, int *__restrict__ sqp_iter_out,
                                                        double *__restrict__ res_eq_out, int *__restrict__ qp_iter_out,
                                                        long long *__restrict__ prof_out, StateIO io)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= B) return;
    const int b = trajectory_of_block(blockIdx.x, B);
    if ((slot_flags(io, b) & ST_KEEP_ITERATE) && io.stopped[slot_of(io, b)]) return;      // this solver's loop has ended: outputs of its last call stand
    const Lds L = carve(smem, d);
    const int N = d.N;
    const double *xi = xinit + (size_t)b * ext_nx(d);
    const double *pb_own = params + (size_t)b * N * d.npar;
    const double *pb = params + (size_t)param_base_of(io, b) * N * d.npar;      // cost / ellipsoid / spline entries: the (possibly shared) row block
    const double slack = d.slack ? xi[NX] : 0.0;              // pinned by x_0 = xinit and slack' = 0 (tmpc_stage.hpp)

    // loadWarmstart (acados_solver_interface.cpp:274-284), or the iterate the handle holds; fresh or kept multipliers
    for (int e = tid; e < (N + 1) * NV; e += NT) {
        const int k = e / NV, i = e - k * NV;
        L.z[e] = (slot_flags(io, b) & ST_KEEP_ITERATE) ? io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] : x0[((size_t)b * (N + 1) + k) * ext_nv(d) + i];
    }
    for (int e = tid; e < (N + 1) * NX; e += NT) L.pi[e] = (slot_flags(io, b) & ST_KEEP_MULTIPLIERS) ? io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] : 0.0;
    for (int r = tid; r < L.nrows; r += NT) {
        double l0 = 0.0;
        if ((slot_flags(io, b) & ST_KEEP_MULTIPLIERS) && r < L.NG) {
            const int j = r % L.nh;
            l0 = ((j < d.n_up) ? 1.0 : -1.0) * io.lamh[(size_t)slot_of(io, b) * L.NG + r];         // lam = -sgn (lam_upper - lam_lower)
        }
        L.lam[r] = l0;
    }
    __syncthreads();
    if (tid < NU) L.z[N * NV + tid] = 0.0;
    __syncthreads();

    Prof pf; pf.init(prof_out);
    const long long t_begin = prof_out ? clock64() : 0;
    int status = 0, qp_status = 0, sqp_iter = 0, qp_iter_total = 0;
    for (int it = 0; it < d.n_sqp; it++) {
        pf.start();
        linearise<false, false, 64, CM>(L, d, tid, pb, slack, pb_own);
        // QP primal start: dz = 0 except dx_0 = xinit - x_0; duals 0
        for (int e = tid; e < (N + 1) * NV; e += NT) L.v[e] = 0.0;
        for (int e = tid; e < (N + 1) * NX; e += NT) L.pq[e] = 0.0;
        __syncthreads();
        if (tid < NX) L.v[NU + tid] = xi[tid] - L.z[NU + tid];
        __syncthreads();
        pf.stop(PH_LIN);
        int iters = 0;
        qp_status = ipm_solve(L, d, tid, &iters, pf);
        sqp_iter = it + 1; qp_iter_total += iters;
        if (qp_status != 0 && qp_status != 2) { status = 4; break; }      // ACADOS_QP_FAILURE, no step
        status = 0;
        __syncthreads();
        for (int e = tid; e < (N + 1) * NV; e += NT) {
            const int k = e / NV, i = e - k * NV;
            if (!(k == N && i < NU)) L.z[e] += L.v[e];
        }
        for (int e = tid; e < N * NX; e += NT) L.pi[NX + e] = L.pq[NX + e];
        __syncthreads();
        if (qp_status != 0) break;
    }

    if (io.flags & ST_STORE) {
        for (int e = tid; e < (N + 1) * NV; e += NT) io.z[(size_t)slot_of(io, b) * (N + 1) * NV + e] = L.z[e];
        for (int e = tid; e < (N + 1) * NX; e += NT) io.pi[(size_t)slot_of(io, b) * (N + 1) * NX + e] = L.pi[e];
        for (int r = tid; r < L.NG; r += NT) io.lamh[(size_t)slot_of(io, b) * L.NG + r] = (((r % L.nh) < d.n_up) ? 1.0 : -1.0) * L.lam[r];
        if (tid == 0) { if (sqp_iter > 0) io.stopped[slot_of(io, b)] = qp_status != 0; io.valid[slot_of(io, b)] = 1; }
    }
    solve_epilogue<CM>(L, d, tid, b, xi, pb, slack, status, qp_status, sqp_iter, qp_iter_total, xtraj, utraj, pobj, exit_code,
                       qp_status_out, sqp_iter_out, res_eq_out, qp_iter_out, prof_out, pf, t_begin);
}

}  // namespace tmpc
#include "tmpc_scan.hpp"
#include "tmpc_fast.hpp"
