// mpc_planner_amd/csrc/tmpc_solve.hip -- batched SQP_RTI solve kernel for gfx950 (MI355X) + C-ABI.
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
// One translation unit: tmpc_stage.hpp (stage functions: dynamics, cost, rows, MIRROR), tmpc_riccati.hpp (Riccati
// factorisation + vector sweeps), tmpc_fast.hpp (the register-resident "fast" solve kernels), tmpc_aux_kernels.hpp
// (selection, records, f-1/f-2/f-3 helper kernels); this file holds the LDS layout, wave helpers, the generic solve
// kernel, kernel dispatch and the C-ABI.
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
#include "tmpc_lanes_api.hpp"

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
// dyn8[N][8] followed by 16 constants; an entry is addressed by a 4-bit code: 0..7 = dyn8 entry of the stage, 8..11 = 0, 1, dt,
// dt^2/2.  The remaining 12 constants are rows psi, v, s of [B A] in dyn8 column order (a, w, psi, v) for the forward sweep.
// Reading [B A] through the table returns exactly the values the dense copy held (zeros and ones included), so every sum that
// runs over a row or column of [B A] keeps its operation order: results are bitwise those of the dense layout.
constexpr int BAC_0 = 8, BAC_1 = 9, BAC_DT = 10, BAC_H = 11, BA_NCONST = 16;
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
                  : ba_pack(BAC_H, BAC_0, BAC_0, BAC_0, BAC_0, BAC_DT, BAC_1);
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
        const double dt = d.dt, h = d.hdt2;
        //                          0    1    dt  h  | psi: a  w   psi  v  | v: a   w    psi  v  | s: a  w    psi  v
        const double c[BA_NCONST] = {0.0, 1.0, dt, h,   0.0, dt, 1.0, 0.0,   dt, 0.0, 0.0, 1.0,   h, 0.0, 0.0, dt};
        double val = 0.0;
#pragma unroll
        for (int i = 0; i < BA_NCONST; i++) if (i == tid) val = c[i];
        tab[d.N * 8 + tid] = val;
    }
}
// doubles of one workgroup's global workspace
__host__ __device__ inline int ws_doubles(int N) { return (N + 1) * NV + (N + 1) * NX + (N + 1) * NP28 + (N + 1) * NV + (N + 1) * NX; }


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

// Workgroup reductions for the two-wave (128-thread) variant of the fast kernel: wave reduction, then one LDS exchange.
// `scr` = L.scr (64 doubles); slots [base, base + 2 * n) are used.  NTH == 64 reduces to the wave reduction.
template <int NTH, typename Op>
__device__ __forceinline__ double blk_combine(double x, double *scr, int tid, int slot, Op op)
{
    if constexpr (NTH == 64) return x;
    // every call site has its own slot and is reached once per interior-point iteration, with barriers in between:
    // the previous readers of the slot are long done
    if ((tid & 63) == 0) scr[slot * 2 + (tid >> 6)] = x;
    __syncthreads();
    return op(scr[slot * 2], scr[slot * 2 + 1]);
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

// The five convergence quantities of an interior-point iteration at once: one LDS exchange and one barrier in the two-wave kernels
// instead of five (same slots, same combination order as five blk_max / blk_sum calls: bitwise the same values).
template <int NTH>
__device__ __forceinline__ void blk_residuals(double &g, double &b, double &dd, double &m, double &mu, double *scr, int tid)
{
    g = wave_max(g); b = wave_max(b); dd = wave_max(dd); m = wave_max(m); mu = wave_sum(mu);
    if constexpr (NTH > 64) {
        if ((tid & 63) == 0) { const int w = tid >> 6; scr[w] = g; scr[2 + w] = b; scr[4 + w] = dd; scr[6 + w] = m; scr[8 + w] = mu; }
        __syncthreads();
        double v[10];
#pragma unroll
        for (int i = 0; i < 10; i++) v[i] = scr[i];
        g = fmax(v[0], v[1]); b = fmax(v[2], v[3]); dd = fmax(v[4], v[5]); m = fmax(v[6], v[7]); mu = v[8] + v[9];
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
    if constexpr (FAST && !CP && NTH == 128) {
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
                double *Dr = L.D + (k * nh + r) * 3;
                const double sg = (r < d.n_up) ? -1.0 : 1.0;     // fast layouts keep the SIGNED row Jacobian (ipm_fast reads it as it is)
                Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy; Dr[2] = sg * ro.gp;
                const double bound = (r < d.n_up) ? 0.0 : 1.0;
                L.beta[k * nh + r] = bound - ro.h;
            }
        };
        // both waves park their share of W (wave 1 in the stage's W slot, wave 0 in the -- idle -- residual arrays of the interior-point
        // work region), so that after the barrier each of them has the complete W and MIRROR can be shared as well: with a zero disc offset
        // W is block diagonal under {a, w, psi, v} | {x, y, spline} (mirror7), and the two blocks are regularised on different waves --
        // bitwise what mirror7 computes.  A coupled W (any cross entry != 0) takes the 7 x 7 iteration on wave 0.
        double *W0s = L.scan + k * NP28;                     // (N * NP28 doubles behind the layout: every two-wave launch allocates them)
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
                mirror_n<4>(Ba, d.reg_eps);
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
    const int k = owner ? tid_l : N - 1;
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
            if (owner) {
                // fast layouts (the register-row kernels) keep the SIGNED row Jacobian sgn D -- upper-bounded rows -1, lower-bounded +1 -- so that
                // the row passes of ipm_fast read their coefficients as they are; the generic kernel keeps D and applies the sign itself
                const double sg = FAST ? ((r < d.n_up) ? -1.0 : 1.0) : 1.0;
                if constexpr (CP) {
                    // packed Jacobians: (gx, gy) for topology rows (gp == 0 exactly, lin_row_eval), triples for the others
                    double *Dr = L.D + k * L.dstride + (r < L.n_pair ? 2 * r : 3 * r - L.n_pair);
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy;
                    if (r >= L.n_pair) Dr[2] = sg * ro.gp;
                } else {
                    double *Dr = L.D + (k * nh + r) * 3;
                    Dr[0] = sg * ro.gx; Dr[1] = sg * ro.gy; Dr[2] = sg * ro.gp;
                }
                const double bound = (r < d.n_up) ? 0.0 : 1.0;
                L.beta[k * nh + r] = bound - ro.h;
            }
        };
        stage_linearise<CM>(d, z, p, 1, L.pi[(k + 1) * NX + 0], L.pi[(k + 1) * NX + 1], lamh, sink, W, g, BA, xn, slack,
                        L.W + k * NP28, own_delta);         // (generated solvers park the cost Hessian in the stage's W slot)
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
        mirror7(W, d.reg_eps);
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
        if constexpr (CM == 1) { CostOutCA co; cost_eval_ca(d, z, pb + (size_t)ks * d.npar, 1, co, false, slack); cval = co.val; }
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
        const int k = e / nxe, i = e - k * nxe;
        xtraj[(size_t)b * (N + 1) * nxe + e] = i < NX ? L.z[k * NV + NU + i] : slack;      // the pinned slack state
    }
    for (int e = tid_o; e < N * NU; e += nth) {
        const int k = e / NU, i = e - k * NU;
        utraj[(size_t)b * N * NU + e] = L.z[k * NV + i];
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
#ifndef TMPC_PROF_TU
template <int CM>      // cost model (Dims::cost_model): 0 MPCC contouring, 1 curvature-aware contouring
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

#endif  // TMPC_PROF_TU

}  // namespace tmpc
#include "tmpc_scan.hpp"
#include "tmpc_fast.hpp"
// Hand-written fast shapes (NLIN, MM, LPS, NTH): the list pick_fast_kernel / pick_latency_kernel dispatch over.  The library build
// splits them over translation units to shorten the build: the profiled twins (PROF = true, tmpc_debug_profile) are compiled in a
// second unit (-DTMPC_PROF_TU: this file up to here + their explicit instantiations), the main unit (-DTMPC_PROF_EXTERN) only declares
// them.  Without either macro (generated solvers, probes) everything is instantiated implicitly in one unit.
#define TMPC_FAST_SHAPES(X) X(8, 8, 4, 128) X(12, 12, 4, 128) X(20, 8, 4, 128) X(-1, 6, 4, 128) X(-1, 9, 4, 128) X(-1, 12, 4, 128) X(0, 4, 3, 64) \
    X(8, 8, 3, 64) X(12, 12, 3, 64) X(24, 0, 3, 64) X(-1, 7, 3, 64) X(-1, 10, 3, 64) X(-1, 13, 3, 64) X(-1, 9, 6, 128) X(0, 4, 2, 64) X(8, 8, 6, 128)
#define TMPC_KARGS tmpc::Dims, int, const double *, const double *, const double *, double *, double *, double *, int *, int *, int *, double *, int *, long long *, tmpc::StateIO
#if defined(TMPC_PROF_TU)
#define TMPC_X(a, b, c, e) template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, true>(TMPC_KARGS);
TMPC_FAST_SHAPES(TMPC_X)
#undef TMPC_X
#ifndef TMPC_GENERATED_STAGE
template __global__ void tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>(TMPC_KARGS);      // profiled twin of latency mode 2 (cfg 2)
#endif
#elif defined(TMPC_PROF_EXTERN)
#define TMPC_X(a, b, c, e) extern template __global__ void tmpc::tmpc_solve_fast_kernel<a, b, c, e, true>(TMPC_KARGS);
TMPC_FAST_SHAPES(TMPC_X)
#undef TMPC_X
#ifndef TMPC_GENERATED_STAGE
extern template __global__ void tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>(TMPC_KARGS);
#endif
#endif
#ifdef TMPC_PROF_TU
#elif defined(TMPC_SINGLE_COMPACT)
template __global__ void tmpc::tmpc_solve_compact_kernel<TMPC_SINGLE_COMPACT>(tmpc::Dims, int, const double *, const double *, const double *,
                                                                             double *, double *, double *, int *, int *, int *, double *, int *,
                                                                             long long *, tmpc::StateIO);
#elif defined(TMPC_SINGLE_KERNEL)
// Experiment builds (tools/kernel_probe.sh): one instantiation only, no C-ABI -- seconds instead of minutes per compile when
// looking at one kernel's registers / ISA.  TMPC_SINGLE_KERNEL = the template argument list, e.g. -DTMPC_SINGLE_KERNEL=8,8,3,64,false
template __global__ void tmpc::tmpc_solve_fast_kernel<TMPC_SINGLE_KERNEL>(tmpc::Dims, int, const double *, const double *, const double *,
                                                                          double *, double *, double *, int *, int *, int *, double *, int *,
                                                                          long long *, tmpc::StateIO);
#else
#include "tmpc_aux_kernels.hpp"

// =================================================================================================
// C-ABI
// =================================================================================================
namespace tmpc {
typedef void (*SolveKernel)(Dims, int, const double *, const double *, const double *, double *, double *, double *, int *,
                            int *, int *, double *, int *, long long *, StateIO);
// Registered fast shapes (upper-bounded rows n_lin + n_slk, ellipsoids M) x lanes-per-stage; anything else runs the generic kernel.
// Only instantiations that compile WITHOUT scratch (zero VGPR spills) are registered: __graft_entry__.build() checks
// the compiler's resource remarks and fails otherwise.  Reason: with > ~100 spilled VGPRs this kernel was observed to
// return wrong iterates (spill/reload around partially-masked regions), see DESIGN.md section 5.  Shapes with more rows
// per lane ((8,8) at 2 lanes/stage for N > 21, (12,12) at 2 lanes/stage) therefore use the generic kernel for now.  The library is
// built with -mllvm -disable-machine-licm: hoisted constant materialisations were what pushed (12,12,3) into scratch.
// prof: the instrumented instantiation (tmpc_debug_profile) instead of the production one.
#define TMPC_FAST(...) (prof ? (SolveKernel)tmpc_solve_fast_kernel<__VA_ARGS__, true> : (SolveKernel)tmpc_solve_fast_kernel<__VA_ARGS__, false>)
static SolveKernel pick_fast_kernel(const Dims &d, int *threads, bool prof)
{
    *threads = NT;
    if (getenv("TMPC_FORCE_GENERIC")) return nullptr;
    const int lps = (3 * d.N <= NT) ? 3 : ((2 * d.N <= NT) ? 2 : 0);
#ifndef TMPC_GENERATED_STAGE
    if (d.cost_model == 1) {
        // curvature-aware contouring (BASELINE configs[2]): the cfg-3 shape on the two-wave kernel, every other row mix of N <= 20 on the
        // runtime-shape one-wave kernel, anything else on the generic kernel -- all instantiated with CM = 1 (no profiled twins)
        if (prof) return nullptr;
        const int nrc = d.n_up + d.M + 14;
        if (lps != 3 && 4 * d.N <= 128 && d.n_up == 20 && d.M == 8 && !getenv("TMPC_NO_TWO_WAVE")) {
            *threads = 128;
            return (SolveKernel)tmpc_solve_fast_kernel<20, 8, 4, 128, false, Solo, 1>;
        }
        if (lps == 3 && nrc <= 3 * 13) return (SolveKernel)tmpc_solve_fast_kernel<-1, 13, 3, 64, false, Solo, 1>;
        return nullptr;
    }
#endif
#ifdef TMPC_GENERATED_STAGE
    // generated solver: one row shape (tmpc_gen::NH upper-bounded rows); the fast instantiations are compiled only when the
    // generator's build found them free of scratch (TMPC_GEN_FAST / TMPC_GEN_FAST2 set by codegen/build.py)
#ifdef TMPC_GEN_FAST
    if (lps == 3) return TMPC_FAST(tmpc_gen::NH, 0, 3, 64);
#endif
#ifdef TMPC_GEN_FAST2
    if (lps != 3 && 4 * d.N <= 128) { *threads = 128; return TMPC_FAST(tmpc_gen::NH, 0, 4, 128); }
#endif
    return nullptr;
#else
    const int nr = d.n_up + d.M + 14;                    // interior-point rows per stage
    if (lps != 3 && 4 * d.N <= 128 && !getenv("TMPC_NO_TWO_WAVE")) {
        // two waves per trajectory, 4 lanes per stage (22 <= N <= 32: the reference's default N = 30 and BASELINE cfg 3)
        SolveKernel k2 = nullptr;
        if (d.n_up == 8 && d.M == 8) k2 = TMPC_FAST(8, 8, 4, 128);
        else if (d.n_up == 12 && d.M == 12) k2 = TMPC_FAST(12, 12, 4, 128);  // mpc_planner_jackalsimulator defaults (N = 30, 12 obstacles)
        else if (d.n_up == 20 && d.M == 8) k2 = TMPC_FAST(20, 8, 4, 128);    // cfg 3: 8 topology + 12 decomp rows + 8 ellipsoids
        else if (nr <= 4 * 6) k2 = TMPC_FAST(-1, 6, 4, 128);                 // any other row mix: runtime-shape instantiations
        else if (nr <= 4 * 9) k2 = TMPC_FAST(-1, 9, 4, 128);                 //   (e.g. mpc_planner_jackal: N = 30, 5 obstacles)
        else if (nr <= 4 * 12) k2 = TMPC_FAST(-1, 12, 4, 128);
        if (k2) { *threads = 128; return k2; }
    }
    if (lps == 3) {
        if (d.n_up == 0 && d.M == 4) return TMPC_FAST(0, 4, 3, 64);
        if (d.n_up == 8 && d.M == 8) return TMPC_FAST(8, 8, 3, 64);
        if (d.n_up == 12 && d.M == 12) return TMPC_FAST(12, 12, 3, 64);      // zero scratch only with machine-LICM off (build flag)
        if (d.n_up == 24 && d.M == 0) return TMPC_FAST(24, 0, 3, 64);        // SH-MPC: 24 scenario halfspaces (cfg 5)
        if (nr <= 3 * 7) return TMPC_FAST(-1, 7, 3, 64);                     // runtime-shape instantiations
        if (nr <= 3 * 10) return TMPC_FAST(-1, 10, 3, 64);
        if (nr <= 3 * 13) return TMPC_FAST(-1, 13, 3, 64);
        if (d.N <= 2 * (64 / 6) && nr <= 6 * 9 && !getenv("TMPC_NO_TWO_WAVE")) {   // more rows: two waves, 6 lanes per stage
            *threads = 128;                                                  //   (mpc_planner_rosnavigation T-MPC: 24 + 12 rows)
            return TMPC_FAST(-1, 9, 6, 128);
        }
    } else if (lps == 2) {
        if (d.n_up == 0 && d.M == 4) return TMPC_FAST(0, 4, 2, 64);
    }
    return nullptr;
#endif
}
// Compact variant (tmpc_fast.hpp: tmpc_solve_compact_kernel): two waves per SIMD, eight trajectories per CU, persistent
// workgroups.  Bitwise the same results as the fast kernel of the shape (tools/ab_compare.py against TMPC_NO_COMPACT=1).
// Round 4: the shapes with 13 rows per lane ((12,12) and (24,0) at three lanes per stage: cfg 4, cfg 5) fit 256 registers too since the
// row passes are specialised by the compile-time kind of each row slot (FastCfg::KIND): 238 registers, zero scratch; their larger row tables
// allow 7 (cfg 4: 23.3 KB) and 6 (cfg 5: 25.2 KB) workgroups per CU.  The runtime-shape instantiation with 13 rows per lane still spills
// (168 B) and is not registered.
static SolveKernel pick_compact_kernel(const Dims &d, bool prof)
{
#ifndef TMPC_GENERATED_STAGE
    if (getenv("TMPC_FORCE_GENERIC") || getenv("TMPC_NO_COMPACT") || prof || d.N > 20 || d.cost_model != 0) return nullptr;
    const int nr = d.n_up + d.M + 14;                    // interior-point rows per stage
    if (d.n_up == 8 && d.M == 8) return (SolveKernel)tmpc_solve_compact_kernel<8, 8, 3, false>;
    if (d.n_up == 0 && d.M == 4) return (SolveKernel)tmpc_solve_compact_kernel<0, 4, 3, false>;
    if (d.n_up == 12 && d.M == 12) return (SolveKernel)tmpc_solve_compact_kernel<12, 12, 3, false>;
    if (d.n_up == 24 && d.M == 0) return (SolveKernel)tmpc_solve_compact_kernel<24, 0, 3, false>;
    if (nr <= 3 * 7) return (SolveKernel)tmpc_solve_compact_kernel<-1, 7, 3, false>;       // runtime-shape instantiations
    if (nr <= 3 * 10) return (SolveKernel)tmpc_solve_compact_kernel<-1, 10, 3, false>;
#endif
    (void)d; (void)prof;
    return nullptr;
}
// Latency variant (tmpc_set_latency_mode): two waves per trajectory at 6 lanes per stage, built for two waves per SIMD
// (<= 256 registers, so four trajectories per CU stay resident).  The stage-parallel phases run on twice the lanes:
// -8 % kernel time on a 64-trajectory control tick; on a saturated GPU the one-wave kernel is as fast or faster, which is
// why it stays the default.  The variant is chosen by the caller, never by the batch size: a trajectory's result does
// not depend on what else is in the launch.
static SolveKernel pick_latency_kernel(const Dims &d, bool prof)
{
#ifndef TMPC_GENERATED_STAGE
    if (getenv("TMPC_FORCE_GENERIC") || getenv("TMPC_NO_TWO_WAVE") || d.N > 2 * (64 / 6) || d.cost_model != 0) return nullptr;
    if (d.n_up == 8 && d.M == 8) return TMPC_FAST(8, 8, 6, 128);
#endif
    (void)d; (void)prof;
    return nullptr;
}
// Latency variant 2 (tmpc_set_latency_mode(h, 2)): one wave per trajectory like the fast kernels, the interior-point Newton systems
// solved parallel in time (tmpc_scan.hpp) instead of by the sequential Riccati recursion.  One workgroup per CU is what a control
// tick gives it anyway: built for one wave per SIMD (all 512 registers, 73 KB of LDS).  Another factorisation of the same systems:
// steps agree with the recursion's to rounding (~1e-6 of a step on ill-conditioned late iterations, like the recursion itself
// against an exact solve), so iteration counts can differ by one where a residual sits at the tolerance -- the caller opts in.
static SolveKernel pick_scan_kernel(const Dims &d, int *threads, int *sl)
{
    *sl = 3;
#ifndef TMPC_GENERATED_STAGE
    if (getenv("TMPC_FORCE_GENERIC") || d.N > 31 || d.N < 2 || d.cost_model != 0) return nullptr;
    if (d.N > 20) {                                              // 21 <= N <= 31 (cfg 3, the reference's N = 30 defaults): two lanes per stage in the
        if (d.n_up + d.M + 14 > 4 * 12) return nullptr;          // Newton solve, the runtime-shape two-wave kernel (4 lanes per stage, up to 34 rows) around it
        *threads = 128; *sl = 2;
        return (SolveKernel)tmpc_solve_fast_kernel<-1, 12, 4, 128, false, ScanSoloT<2>>;
    }
    const char *w = getenv("TMPC_SCAN_WAVES");               // A/B: "1" = one wave per trajectory
    if (d.n_up == 8 && d.M == 8 && d.N <= 2 * (64 / 6) && !(w && atoi(w) == 1)) { *threads = 128; return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 6, 128, false, ScanSolo>; }
    if (d.n_up == 8 && d.M == 8) { *threads = 64; return (SolveKernel)tmpc_solve_fast_kernel<8, 8, 3, 64, false, ScanSolo>; }
    if (d.N <= 2 * (64 / 6) && d.n_up + d.M + 14 <= 6 * 9) {     // every other row mix of the one-wave shapes (cfg 1, cfg 4, cfg 5, ...): runtime row counts, two waves
        *threads = 128;
        return (SolveKernel)tmpc_solve_fast_kernel<-1, 9, 6, 128, false, ScanSolo>;
    }
#endif
    (void)d; (void)threads;
    return nullptr;
}
}  // namespace tmpc

struct tmpc_handle {
    tmpc::Dims d;
    int B_max = 0, B = 0, device = 0;
    hipStream_t stream = nullptr;
    // inputs: owned staging buffers (tmpc_set_batch) or borrowed device pointers (tmpc_set_batch_device)
    double *o_xinit = nullptr, *o_x0 = nullptr, *o_params = nullptr;
    const double *xinit = nullptr, *x0 = nullptr, *params = nullptr;
    double *xtraj = nullptr, *utraj = nullptr, *pobj = nullptr, *res_eq = nullptr, *d_weight = nullptr;
    int *exit_code = nullptr, *qp_status = nullptr, *sqp_iter = nullptr, *qp_iter = nullptr, *d_best = nullptr;
    uint8_t *d_disabled = nullptr;
    size_t lds_bytes = 0;
    tmpc::SolveKernel kernel = nullptr;
    int threads = tmpc::NT;          // threads per trajectory (64, or 128 for the two-wave fast variant)
    tmpc::SolveKernel kernel_lat = nullptr;   // optional latency variant (128 threads), used when latency_mode is 1
    tmpc::SolveKernel kernel_scan = nullptr;  // optional latency variant 2 (parallel-in-time Newton solve, 64 threads)
    size_t lds_bytes_scan = 0;
    int scan_threads = 64, scan_sl = 3;
    size_t lds_bytes_fast = 0;                // LDS of the fast-layout kernels (the profiled twin) when `kernel` is compact
    size_t lds_bytes_fast2 = 0;               // ... of their two-wave variants (kernel_lat): + the W shares parked during the linearisation
    bool compact = false;                     // `kernel` is a compact persistent kernel: grid = resident workgroups, needs ws + ticket
    int grid_max = 0;                         // resident workgroups of the compact kernel on this device
    double *ws = nullptr;                     // [grid_max][ws_doubles(N)] per-workgroup NLP workspace
    int *ticket = nullptr;
    int latency_mode = 0;                     // 0: throughput kernels, 1: two-wave variant, 2: parallel-in-time variant
    bool throughput_mode = false;             // lane-per-trajectory kernels (tmpc_lanes.hip) instead of one wave per trajectory
    tmpc::lanes::Context *lanes = nullptr;    // their HBM workspace, created when the mode is first enabled
    bool fast = false;
    // persistent per-slot solver state (tmpc_solve_iterations), allocated on first use
    double *st_z = nullptr, *st_pi = nullptr, *st_lamh = nullptr;
    int *st_stopped = nullptr;
    int *st_has = nullptr;           // [B_max] the slot holds state of an earlier tmpc_solve_iterations (set by the kernels' store)
    int *d_slot = nullptr;           // [B_max] state slot of every batch entry (tmpc_set_slots)
    bool slots_set = false;
    int slots_B = 0;                 // batch size the slot map was given for: a map of another size is refused, never read past its end
    int *d_share = nullptr;          // [B_max] tmpc_set_param_sharing
    int share_B = 0;                 // batch size the sharing map was given for (0: none)
    bool st_valid = false;           // lane kernels (state = their workspace, per launch): it holds the result of a previous call ...
    int st_B = 0;                    // ... for slots [0, st_B)
    // SH-MPC bookkeeping: the sample behind each scenario row of the last tmpc_scenario_halfspaces (i32 [B][N][scn_rows])
    unsigned char *scn_discard = nullptr;     // [B_max][scn_discard_S] scenarios discarded for each trajectory (tmpc_scenario_discard); applies to the next tmpc_scenario_halfspaces
    int scn_discard_S = 0, scn_discard_B = 0, scn_discard_n = 0;
    size_t scn_discard_cap = 0;
    int *scn_sample = nullptr;
    size_t scn_cap = 0;
    int scn_rows = 0, scn_B = 0;
    std::vector<hipEvent_t> ev;      // per-launch timing events (pairs)
    int ev_used = 0;
    bool timing = false;
    std::string err;
};

#define TMPC_HIP_CHECK(h, expr)                                                                     \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return TMPC_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

namespace {
// Scratch device buffers / events of the diagnostic entry points: released on every return path.
struct DevBufs {
    std::vector<void *> p;
    ~DevBufs() { for (void *q : p) if (q) (void)hipFree(q); }
    hipError_t alloc(double **out, size_t bytes) { hipError_t e = hipMalloc(out, bytes ? bytes : 8); if (e == hipSuccess) p.push_back(*out); return e; }
};
struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() { for (auto &e : ev) if (e) (void)hipEventDestroy(e); }
};
}  // namespace

extern "C" {

void tmpc_default_dims(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M) { tmpc_default_dims_ex(d, N, S, n_lin, M, 0, 0); }

void tmpc_default_dims_ex(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M, int32_t n_slk, int32_t slack)
{
    memset(d, 0, sizeof *d);
#ifdef TMPC_GENERATED_STAGE
    // generated solver: the row / parameter structure is fixed by the generated stage functions
    (void)n_lin; (void)M; (void)n_slk; (void)slack;
    n_lin = tmpc_gen::NH; M = 0; n_slk = 0; slack = tmpc_gen::SLACK;
#endif
    d->N = N; d->S = S; d->n_lin = n_lin; d->M = M; d->n_slk = n_slk; d->slack = slack ? 1 : 0;
    tmpc::Dims t; t.S = S; t.n_lin = n_lin; t.M = M; t.n_slk = n_slk; t.slack = d->slack;
    d->npar = tmpc::expected_npar(t);
    d->n_sqp = 10; d->qp_iter_max = 50; d->erk_steps = 3;
    d->dt = 0.2; d->qp_tol = 1e-5; d->reg_eps = 1e-4; d->ipm_mu0 = 0.01; d->ipm_thr0 = 0.01;
    const double lb[TMPC_NV] = {-2.0, -0.8, -2000.0, -2000.0, -M_PI * 4, -0.01, -1.0};
    const double ub[TMPC_NV] = {2.0, 0.8, 2000.0, 2000.0, M_PI * 4, 3.0, 10000.0};
    for (int i = 0; i < TMPC_NV; i++) { d->lb[i] = lb[i]; d->ub[i] = ub[i]; }
}

int tmpc_create(tmpc_handle **out, const tmpc_dims *dims, int32_t B_max, int32_t device)
{
    if (!out || !dims || B_max <= 0) return TMPC_ERR_INVALID;
    *out = nullptr;
    {
        tmpc::Dims t; t.S = dims->S; t.n_lin = dims->n_lin; t.M = dims->M; t.n_slk = dims->n_slk; t.slack = dims->slack;
        if (dims->N < 2 || dims->N > 62 || dims->S < 1 || dims->M < 0 || dims->n_lin < 0 || dims->n_slk < 0 ||
            (dims->slack != 0 && dims->slack != 1) || dims->npar != tmpc::expected_npar(t) || dims->erk_steps < 1 ||
            dims->n_sqp < 1 || dims->qp_iter_max < 1 || !(dims->dt > 0.0) || !(dims->qp_tol > 0.0) || !(dims->reg_eps > 0.0) ||
            !(dims->ipm_mu0 > 0.0) || !(dims->ipm_thr0 > 0.0) || (dims->cost_model != 0 && dims->cost_model != 1))
            return TMPC_ERR_INVALID;
#ifdef TMPC_GENERATED_STAGE
        if (dims->n_lin != tmpc_gen::NH || dims->M != 0 || dims->n_slk != 0 || dims->slack != tmpc_gen::SLACK || dims->cost_model != 0) return TMPC_ERR_INVALID;
#endif
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TMPC_ERR_NO_DEVICE;
    tmpc_handle *h = new tmpc_handle();
    h->device = device; h->B_max = B_max;
    tmpc::Dims &d = h->d;
    d.N = dims->N; d.S = dims->S; d.n_lin = dims->n_lin; d.M = dims->M; d.npar = dims->npar;
    d.n_slk = dims->n_slk; d.slack = dims->slack; d.cost_model = dims->cost_model;
    d.n_sqp = dims->n_sqp; d.qp_iter_max = dims->qp_iter_max; d.erk_steps = dims->erk_steps;
    d.dt = dims->dt; d.qp_tol = dims->qp_tol; d.reg_eps = dims->reg_eps; d.mu0 = dims->ipm_mu0; d.thr0 = dims->ipm_thr0;
    for (int i = 0; i < TMPC_NV; i++) { d.lb[i] = dims->lb[i]; d.ub[i] = dims->ub[i]; }
    tmpc::derive_dims(d);
    h->kernel = tmpc::pick_fast_kernel(d, &h->threads, false);
    if (const char *lm = getenv("TMPC_LATENCY_MODE")) {      // experiments: latency variant regardless of the caller ("0", "1" or "2"; anything else is ignored)
        if ((lm[0] == '0' || lm[0] == '1' || lm[0] == '2') && lm[1] == '\0') h->latency_mode = lm[0] - '0';
    }
    h->fast = h->kernel != nullptr;
    if (h->fast) h->lds_bytes = sizeof(double) * (size_t)tmpc::lds_doubles_fast(d.N, d.n_up + d.M);
    else { h->kernel = d.cost_model == 1 ? tmpc::tmpc_solve_kernel<1> : tmpc::tmpc_solve_kernel<0>; h->lds_bytes = sizeof(double) * (size_t)tmpc::lds_doubles(d.N, d.n_up + d.M); }
    h->lds_bytes_fast = h->lds_bytes;
    // two-wave (128-thread) fast kernels park one share of W per stage behind the layout while they linearise (linearise<.., 128>)
    h->lds_bytes_fast2 = h->lds_bytes_fast + sizeof(double) * (size_t)d.N * tmpc::NP28;
    if (h->fast && h->threads == 128) h->lds_bytes = h->lds_bytes_fast2;
    auto fail = [&](int code) { delete h; return code; };
    if (hipSetDevice(device) != hipSuccess) return fail(TMPC_ERR_HIP);
    if (h->lds_bytes > 160 * 1024) return fail(TMPC_ERR_INVALID);
    if (h->fast && h->threads == tmpc::NT && (h->kernel_lat = tmpc::pick_latency_kernel(d, false)) != nullptr) {
        if (hipFuncSetAttribute((const void *)h->kernel_lat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_fast2) != hipSuccess)
            h->kernel_lat = nullptr;
    }
    if (h->fast && (h->threads == tmpc::NT || d.N > 20) && (h->kernel_scan = tmpc::pick_scan_kernel(d, &h->scan_threads, &h->scan_sl)) != nullptr) {
        h->lds_bytes_scan = h->lds_bytes_fast2 + sizeof(double) * (size_t)(h->scan_sl == 3 ? tmpc::scan::lds_doubles<3>(d.N) : tmpc::scan::lds_doubles<2>(d.N));
        if (h->lds_bytes_scan > 160 * 1024) h->kernel_scan = nullptr;
    }
    if (h->kernel_scan) {
        if (hipFuncSetAttribute((const void *)h->kernel_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_scan) != hipSuccess)
            h->kernel_scan = nullptr;
    }
    if (tmpc::SolveKernel kc = (h->fast && h->threads == tmpc::NT) ? tmpc::pick_compact_kernel(d, false) : nullptr) {
        h->kernel = kc; h->compact = true;
        h->lds_bytes = sizeof(double) * (size_t)tmpc::lds_doubles_compact(d.N, d.n_lin, d.n_up + d.M);
    }
    if (hipFuncSetAttribute((const void *)h->kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h->lds_bytes) != hipSuccess)
        return fail(TMPC_ERR_NO_DEVICE);
    if (h->compact) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)h->kernel, 64, h->lds_bytes) != hipSuccess || per_cu <= 0 ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0)
            return fail(TMPC_ERR_HIP);
        if (const char *e = getenv("TMPC_COMPACT_PER_CU")) { const int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }   // experiments
        h->grid_max = per_cu * cus;
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(TMPC_ERR_HIP);
    const size_t N = d.N, B = B_max;
    bool ok = true;
    const size_t nxe = tmpc::ext_nx(d), nve = tmpc::ext_nv(d);
    ok &= hipMalloc(&h->o_xinit, B * nxe * 8) == hipSuccess;
    ok &= hipMalloc(&h->o_x0, B * (N + 1) * nve * 8) == hipSuccess;
    ok &= hipMalloc(&h->o_params, B * N * d.npar * 8) == hipSuccess;
    ok &= hipMalloc(&h->xtraj, B * (N + 1) * nxe * 8) == hipSuccess;
    ok &= hipMalloc(&h->utraj, B * N * tmpc::NU * 8) == hipSuccess;
    ok &= hipMalloc(&h->pobj, B * 8) == hipSuccess;
    ok &= hipMalloc(&h->res_eq, B * 8) == hipSuccess;
    ok &= hipMalloc(&h->d_weight, B * 8) == hipSuccess;
    ok &= hipMalloc(&h->exit_code, B * 4) == hipSuccess;
    ok &= hipMalloc(&h->qp_status, B * 4) == hipSuccess;
    ok &= hipMalloc(&h->sqp_iter, B * 4) == hipSuccess;
    ok &= hipMalloc(&h->qp_iter, B * 4) == hipSuccess;
    ok &= hipMalloc(&h->d_best, 4) == hipSuccess;
    ok &= hipMalloc(&h->d_disabled, B) == hipSuccess;
    if (h->compact) {
        ok &= hipMalloc(&h->ws, (size_t)h->grid_max * tmpc::ws_doubles(d.N) * 8) == hipSuccess;
        ok &= hipMalloc(&h->ticket, 8 * 4) == hipSuccess;         // one work counter per XCD (next_trajectory)
    }
    if (!ok) { tmpc_destroy(h); return TMPC_ERR_HIP; }
    *out = h;
    return TMPC_OK;
}

void tmpc_destroy(tmpc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->o_xinit, h->o_x0, h->o_params, h->xtraj, h->utraj, h->pobj, h->res_eq, h->d_weight,
                    h->exit_code, h->qp_status, h->sqp_iter, h->qp_iter, h->d_best, h->d_disabled,
                    h->st_z, h->st_pi, h->st_lamh, h->st_stopped, h->st_has, h->d_slot, h->d_share, h->scn_sample, h->scn_discard, h->ws, h->ticket};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    tmpc::lanes::destroy(h->lanes);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *tmpc_last_error(const tmpc_handle *h) { return h ? h->err.c_str() : "null handle"; }

int tmpc_set_batch(tmpc_handle *h, int32_t B, const double *xinit, const double *x0, const double *params)
{
    if (!h || B <= 0 || B > h->B_max || !xinit || !x0 || !params) { if (h) h->err = "tmpc_set_batch: bad argument"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t N = h->d.N;
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_xinit, xinit, (size_t)B * tmpc::ext_nx(h->d) * 8, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_x0, x0, (size_t)B * (N + 1) * tmpc::ext_nv(h->d) * 8, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->o_params, params, (size_t)B * N * h->d.npar * 8, hipMemcpyHostToDevice, h->stream));
    h->xinit = h->o_xinit; h->x0 = h->o_x0; h->params = h->o_params; h->B = B;
    h->scn_B = 0;                      // new parameter rows: the scenario-row bookkeeping of the previous batch no longer describes them
    h->scn_discard_B = 0;
    h->share_B = 0;                    // ... nor does a parameter-sharing map given for them
    return TMPC_OK;
}

int tmpc_set_batch_device(tmpc_handle *h, int32_t B, const void *d_xinit, const void *d_x0, const void *d_params)
{
    if (!h || B <= 0 || B > h->B_max || !d_xinit || !d_x0 || !d_params) { if (h) h->err = "tmpc_set_batch_device: bad argument"; return TMPC_ERR_INVALID; }
    h->xinit = (const double *)d_xinit; h->x0 = (const double *)d_x0; h->params = (const double *)d_params; h->B = B;
    h->scn_B = 0; h->scn_discard_B = 0; h->share_B = 0;
    return TMPC_OK;
}

// One launch over the current batch: n_iter RTI iterations per trajectory + completeOneIteration.  st_flags: ST_* (0 = fresh
// solver instances from the batch's warm start, nothing kept or stored: Solver::solve() of a new capsule).
static int launch_solve(tmpc_handle *h, int n_iter, int st_flags)
{
    const bool rec = h->timing && h->ev_used + 2 <= (int)h->ev.size();
    if (rec) TMPC_HIP_CHECK(h, hipEventRecord(h->ev[h->ev_used], h->stream));
    if (h->throughput_mode) {
        // lane-per-trajectory variant: transpose the reference-layout inputs into the lane-major workspace, then one launch of
        // the scalar-per-lane SQP_RTI program; the workspace itself is the persistent state
        if (tmpc::lanes::stage_in(h->lanes, h->stream, h->B, h->xinit, h->x0, h->params, !(st_flags & tmpc::ST_KEEP_ITERATE),
                                  !(st_flags & tmpc::ST_KEEP_MULTIPLIERS), h->err)) return TMPC_ERR_HIP;
        if (tmpc::lanes::solve(h->lanes, h->stream, h->B, n_iter, (st_flags & tmpc::ST_STORE) != 0, (st_flags & tmpc::ST_COMPLETE) != 0,
                               h->xtraj, h->utraj, h->pobj,
                               h->exit_code, h->qp_status, h->sqp_iter, h->res_eq, h->qp_iter, h->err)) return TMPC_ERR_HIP;
    } else {
        tmpc::Dims dd = h->d;
        dd.n_sqp = n_iter;
        tmpc::StateIO io{h->st_z, h->st_pi, h->st_lamh, h->st_stopped, st_flags, h->ws, h->ticket, (h->slots_set && h->slots_B == h->B) ? h->d_slot : nullptr, h->st_has,
                         (h->share_B == h->B) ? h->d_share : nullptr};      // (a map given for another batch size is not applied)
        const bool lat2 = h->kernel_scan && h->latency_mode == 2;
        const bool lat = !lat2 && h->kernel_lat && h->latency_mode != 0;          // (mode 2 without a scan variant falls back to the two-wave variant)
        const bool cp = h->compact && !lat && !lat2;
        if (cp) TMPC_HIP_CHECK(h, hipMemsetAsync(h->ticket, 0, 8 * 4, h->stream));    // the persistent launch's work counters (one per XCD)
        hipLaunchKernelGGL(lat2 ? h->kernel_scan : lat ? h->kernel_lat : h->kernel, dim3(cp ? (h->B < h->grid_max ? h->B : h->grid_max) : h->B),   // (persistent launch: at most the resident workgroups)
                           dim3(lat2 ? h->scan_threads : lat ? 128 : (cp ? 64 : h->threads)), lat2 ? h->lds_bytes_scan : lat ? h->lds_bytes_fast2 : h->lds_bytes, h->stream, dd, h->B,
                           h->xinit, h->x0, h->params, h->xtraj, h->utraj, h->pobj, h->exit_code, h->qp_status,
                           h->sqp_iter, h->res_eq, h->qp_iter, (long long *)nullptr, io);
        TMPC_HIP_CHECK(h, hipGetLastError());
        if (st_flags & tmpc::ST_COMPLETE) {
            // a failed solve resets the reference's capsule (Solver_acados_reset, acados_solver_interface.cpp:187-191): zero multipliers
            const int n_pi = (h->d.N + 1) * tmpc::NX, n_lam = h->d.N * (h->d.n_up + h->d.M);
            hipLaunchKernelGGL(tmpc::tmpc_state_finalize_kernel, dim3(h->B), dim3(64), 0, h->stream, n_pi, n_lam, h->exit_code, h->st_pi, h->st_lamh,
                               (h->slots_set && h->slots_B == h->B) ? h->d_slot : nullptr);
            TMPC_HIP_CHECK(h, hipGetLastError());
        }
    }
    if (rec) { TMPC_HIP_CHECK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream)); h->ev_used += 2; }
    return TMPC_OK;
}

// the slots' persistent state no longer describes what the handle last solved
static int invalidate_state(tmpc_handle *h)
{
    h->st_valid = false; h->st_B = 0;
    if (h->st_has) TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_has, 0, (size_t)h->B_max * 4, h->stream));
    return TMPC_OK;
}

int tmpc_solve(tmpc_handle *h)
{
    if (!h || h->B <= 0 || !h->xinit) { if (h) h->err = "tmpc_solve: no batch set"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (int rc = invalidate_state(h)) return rc;
    return launch_solve(h, h->d.n_sqp, 0);
}

int tmpc_solve_iterations(tmpc_handle *h, int32_t n_iter, int32_t flags)
{
    if (!h || h->B <= 0 || !h->xinit || n_iter < 0 || (flags & ~15)) { if (h) h->err = "tmpc_solve_iterations: no batch set / bad argument"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->throughput_mode && h->slots_set) { h->err = "tmpc_solve_iterations: slot maps (tmpc_set_slots) are not available with the lane kernels"; return TMPC_ERR_INVALID; }
    if (h->slots_set && h->slots_B != h->B) {
        // the map has one entry per batch entry: entries [slots_B, B) of a larger batch would be read from uninitialised memory
        h->err = "tmpc_solve_iterations: the slot map was given for a batch of " + std::to_string(h->slots_B) + " entries, the current batch has " +
                 std::to_string(h->B) + ": call tmpc_set_slots again after tmpc_set_batch (or clear it with a null map)";
        return TMPC_ERR_INVALID;
    }
    if (!h->throughput_mode && !h->st_z) {
        const size_t B = h->B_max, N = h->d.N, nh = h->d.n_up + h->d.M;
        const size_t sz[5] = {B * (N + 1) * tmpc::NV * 8, B * (N + 1) * tmpc::NX * 8, (B * N * nh + 1) * 8, B * 4, B * 4};
        void **dst[5] = {(void **)&h->st_z, (void **)&h->st_pi, (void **)&h->st_lamh, (void **)&h->st_stopped, (void **)&h->st_has};
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) ok = hipMalloc(dst[i], sz[i]) == hipSuccess && hipMemsetAsync(*dst[i], 0, sz[i], h->stream) == hipSuccess;
        if (!ok) {                          // all or nothing: a later call must not find half of the arrays
            for (int i = 0; i < 5; i++) { if (*dst[i]) (void)hipFree(*dst[i]); *dst[i] = nullptr; }
            h->err = "tmpc_solve_iterations: state allocation failed"; return TMPC_ERR_HIP;
        }
        h->st_valid = false; h->st_B = 0;
    }
    int st = tmpc::ST_STORE;
    if (h->throughput_mode) {
        // lane kernels keep their state per launch, not per slot: nothing to keep on the first call, and a grown batch starts fresh
        if (!h->st_valid) h->st_B = 0;
        if (h->B > h->st_B) h->st_valid = false;
        if (h->st_valid) {
            if (flags & TMPC_ITER_KEEP_ITERATE) st |= tmpc::ST_KEEP_ITERATE;
            if (flags & TMPC_ITER_KEEP_MULTIPLIERS) st |= tmpc::ST_KEEP_MULTIPLIERS;
        }
    } else {
        // wave kernels: the keep-flags apply per slot -- a slot without stored state (first call, grown batch, new slot of a map)
        // starts like a fresh capsule (slot_flags in the kernels)
        if (flags & TMPC_ITER_KEEP_ITERATE) st |= tmpc::ST_KEEP_ITERATE;
        if (flags & TMPC_ITER_KEEP_MULTIPLIERS) st |= tmpc::ST_KEEP_MULTIPLIERS;
    }
    if (flags & TMPC_ITER_COMPLETE) st |= tmpc::ST_COMPLETE;
    // a new solve() of the slots' Solvers: the "iteration loop has ended" marks belong to the previous solve (:105-106 is local to one solve())
    if ((flags & TMPC_ITER_NEW_SOLVE) && !h->throughput_mode && h->st_stopped) TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_stopped, 0, (size_t)h->B_max * 4, h->stream));
    if ((flags & TMPC_ITER_NEW_SOLVE) && h->throughput_mode && h->lanes && tmpc::lanes::clear_stopped(h->lanes, h->stream, h->B_max, h->err)) return TMPC_ERR_HIP;
    const int rc = launch_solve(h, n_iter, st);
    if (rc == TMPC_OK && h->throughput_mode) { h->st_valid = true; if (h->B > h->st_B) h->st_B = h->B; }
    return rc;
}

int tmpc_reset_multipliers(tmpc_handle *h)
{
    if (!h) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (h->throughput_mode) {
        if (h->lanes && tmpc::lanes::reset_multipliers(h->lanes, h->stream, h->B_max, h->err)) return TMPC_ERR_HIP;
    } else if (h->st_pi) {
        const size_t B = h->B_max, N = h->d.N, nh = h->d.n_up + h->d.M;
        TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_pi, 0, B * (N + 1) * tmpc::NX * 8, h->stream));
        TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_lamh, 0, B * N * nh * 8, h->stream));
    }
    return TMPC_OK;
}

int tmpc_set_latency_mode(tmpc_handle *h, int32_t on)
{
    if (!h) return TMPC_ERR_INVALID;
    if (on < 0 || on > 2) return TMPC_ERR_INVALID;
    h->latency_mode = on;
    if (on == 2) return h->kernel_scan ? TMPC_OK : 1;              // 1: accepted, but this shape has no such variant (mode 2 then runs as mode 1 if that exists)
    return (on == 1 && !h->kernel_lat) ? 1 : TMPC_OK;
}

int tmpc_set_slots(tmpc_handle *h, const int32_t *slots)
{
    if (!h) return TMPC_ERR_INVALID;
    if (!slots) { h->slots_set = false; h->slots_B = 0; return TMPC_OK; }
    if (h->B <= 0) { h->err = "tmpc_set_slots: set the batch first (the map has one entry per batch entry)"; return TMPC_ERR_INVALID; }
    std::vector<char> seen((size_t)h->B_max, 0);
    for (int b = 0; b < h->B; b++) {
        if (slots[b] < 0 || slots[b] >= h->B_max || seen[slots[b]]) { h->err = "tmpc_set_slots: slots must be distinct and in [0, B_max)"; return TMPC_ERR_INVALID; }
        seen[slots[b]] = 1;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (!h->d_slot) TMPC_HIP_CHECK(h, hipMalloc(&h->d_slot, (size_t)h->B_max * 4));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_slot, slots, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));            // (the caller's array may go away)
    h->slots_set = true; h->slots_B = h->B;
    return TMPC_OK;
}

int tmpc_set_param_sharing(tmpc_handle *h, const int32_t *base_of)
{
    if (!h) return TMPC_ERR_INVALID;
    if (!base_of) { h->share_B = 0; return TMPC_OK; }
#ifdef TMPC_GENERATED_STAGE
    // generated stage functions read every parameter -- halfspace rows included -- from ONE row block (tmpc_gen::rows has no notion of
    // "own" rows), so the hint cannot be honoured: it is accepted and ignored, as include/tmpc_hip.h says
    h->share_B = 0;
    return TMPC_OK;
#endif
    if (h->B <= 0) { h->err = "tmpc_set_param_sharing: set the batch first (the map has one entry per batch entry)"; return TMPC_ERR_INVALID; }
    for (int b = 0; b < h->B; b++)
        if (base_of[b] < 0 || base_of[b] >= h->B) { h->err = "tmpc_set_param_sharing: entries must be batch indices in [0, B)"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (!h->d_share) TMPC_HIP_CHECK(h, hipMalloc(&h->d_share, (size_t)h->B_max * 4));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_share, base_of, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));            // (the caller's array may go away)
    h->share_B = h->B;
    return TMPC_OK;
}

int tmpc_copy_state(tmpc_handle *dst, tmpc_handle *src)
{
    if (!dst || !src || dst == src) return TMPC_ERR_INVALID;
    const tmpc::Dims &a = dst->d, &b = src->d;
    if (a.N != b.N || a.n_up != b.n_up || a.M != b.M || dst->device != src->device || dst->throughput_mode || src->throughput_mode) {
        dst->err = "tmpc_copy_state: handles of different shape / device / kernel family"; return TMPC_ERR_INVALID;
    }
    if (!src->st_z) return TMPC_OK;                                 // nothing stored yet
    TMPC_HIP_CHECK(dst, hipSetDevice(dst->device));
    TMPC_HIP_CHECK(dst, hipStreamSynchronize(src->stream));
    if (!dst->st_z) {                                               // allocate through the regular path: an evaluation-only call on the (unset) batch is not possible, so inline it
        const size_t B = dst->B_max, N = a.N, nh = a.n_up + a.M;
        const size_t sz[5] = {B * (N + 1) * tmpc::NV * 8, B * (N + 1) * tmpc::NX * 8, (B * N * nh + 1) * 8, B * 4, B * 4};
        void **p[5] = {(void **)&dst->st_z, (void **)&dst->st_pi, (void **)&dst->st_lamh, (void **)&dst->st_stopped, (void **)&dst->st_has};
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) ok = hipMalloc(p[i], sz[i]) == hipSuccess && hipMemsetAsync(*p[i], 0, sz[i], dst->stream) == hipSuccess;
        if (!ok) { for (int i = 0; i < 5; i++) { if (*p[i]) (void)hipFree(*p[i]); *p[i] = nullptr; } dst->err = "tmpc_copy_state: allocation failed"; return TMPC_ERR_HIP; }
    }
    const size_t n = (size_t)(dst->B_max < src->B_max ? dst->B_max : src->B_max), N = a.N, nh = a.n_up + a.M;
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_z, src->st_z, n * (N + 1) * tmpc::NV * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_pi, src->st_pi, n * (N + 1) * tmpc::NX * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_lamh, src->st_lamh, n * N * nh * 8, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_stopped, src->st_stopped, n * 4, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipMemcpyAsync(dst->st_has, src->st_has, n * 4, hipMemcpyDeviceToDevice, dst->stream));
    TMPC_HIP_CHECK(dst, hipStreamSynchronize(dst->stream));
    return TMPC_OK;
}

int tmpc_clear_slot(tmpc_handle *h, int32_t slot)
{
    if (!h || slot < 0 || slot >= h->B_max) { if (h) h->err = "tmpc_clear_slot: slot out of range"; return TMPC_ERR_INVALID; }
    if (h->throughput_mode) { h->err = "tmpc_clear_slot: the lane kernels keep their state per launch, not per slot"; return TMPC_ERR_INVALID; }
    if (!h->st_has) return TMPC_OK;                                 // nothing stored yet: every slot is fresh
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_has + slot, 0, 4, h->stream));
    TMPC_HIP_CHECK(h, hipMemsetAsync(h->st_stopped + slot, 0, 4, h->stream));
    return TMPC_OK;
}

int tmpc_set_throughput_mode(tmpc_handle *h, int32_t on)
{
    if (!h) return TMPC_ERR_INVALID;
    if (on && h->d.cost_model != 0) { h->err = "tmpc_set_throughput_mode: the lane kernels have the MPCC contouring cost only"; return TMPC_ERR_INVALID; }
    if (on && !h->lanes) {
        TMPC_HIP_CHECK(h, hipSetDevice(h->device));
        h->lanes = tmpc::lanes::create(h->d, h->B_max, h->err);
        if (!h->lanes) return TMPC_ERR_HIP;
    }
    if (h->throughput_mode != (on != 0)) { if (int rc = invalidate_state(h)) return rc; }      // the two kernel families keep their persistent state separately
    h->throughput_mode = on != 0;
    return TMPC_OK;
}

int tmpc_synchronize(tmpc_handle *h)
{
    if (!h) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_get(tmpc_handle *h, double *xtraj, double *utraj, double *pobj, int32_t *exit_code, int32_t *qp_status,
             int32_t *sqp_iter, double *res_eq, int32_t *qp_iter_total)
{
    if (!h || h->B <= 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t N = h->d.N, B = h->B;
    auto cp = [&](void *dst, const void *src, size_t n) { return dst ? hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, h->stream) : hipSuccess; };
    TMPC_HIP_CHECK(h, cp(xtraj, h->xtraj, B * (N + 1) * tmpc::ext_nx(h->d) * 8));
    TMPC_HIP_CHECK(h, cp(utraj, h->utraj, B * N * tmpc::NU * 8));
    TMPC_HIP_CHECK(h, cp(pobj, h->pobj, B * 8));
    TMPC_HIP_CHECK(h, cp(res_eq, h->res_eq, B * 8));
    TMPC_HIP_CHECK(h, cp(exit_code, h->exit_code, B * 4));
    TMPC_HIP_CHECK(h, cp(qp_status, h->qp_status, B * 4));
    TMPC_HIP_CHECK(h, cp(sqp_iter, h->sqp_iter, B * 4));
    TMPC_HIP_CHECK(h, cp(qp_iter_total, h->qp_iter, B * 4));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_select_best(tmpc_handle *h, int32_t first, int32_t count, const double *weight, const uint8_t *disabled, int32_t *best)
{
    if (!h || !best || first < 0 || count <= 0 || first + count > h->B) { if (h) h->err = "tmpc_select_best: bad range"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    if (weight) TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_weight, weight, (size_t)count * 8, hipMemcpyHostToDevice, h->stream));
    if (disabled) TMPC_HIP_CHECK(h, hipMemcpyAsync(h->d_disabled, disabled, (size_t)count, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(tmpc::tmpc_select_best_kernel, dim3(1), dim3(256), 0, h->stream, first, count, h->pobj, h->exit_code,
                       weight ? h->d_weight : nullptr, disabled ? h->d_disabled : nullptr, h->d_best);
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipMemcpyAsync(best, h->d_best, 4, hipMemcpyDeviceToHost, h->stream));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return TMPC_OK;
}

int tmpc_get_stream(tmpc_handle *h, void **stream)
{
    if (!h || !stream) return TMPC_ERR_INVALID;
    *stream = (void *)h->stream;
    return TMPC_OK;
}

int tmpc_kernel_info(const tmpc_handle *h, char *buf, int32_t capacity)
{
    if (!h || !buf || capacity <= 0) return TMPC_ERR_INVALID;
    const char *family = h->throughput_mode ? "lanes (one lane per trajectory)"
                         : !h->fast        ? "generic (one wave per trajectory, rows in LDS)"
                         : !h->compact     ? (h->threads == 128 ? "fast, two waves per trajectory" : "fast (one wave per trajectory)")
                                           : "compact (one wave per trajectory, two waves per SIMD)";
    const int n = snprintf(buf, (size_t)capacity, "%s; trajectories per workgroup %d; LDS %zu B per workgroup; %s", family, 1,
                           h->lds_bytes, h->compact ? (std::string("persistent launch, resident workgroups ") + std::to_string(h->grid_max)).c_str()
                                                    : "one workgroup per trajectory");
    return n < capacity ? n : capacity - 1;
}

int tmpc_result_device_ptrs(tmpc_handle *h, void **d_pobj, void **d_exit_code)
{
    if (!h) return TMPC_ERR_INVALID;
    if (d_pobj) *d_pobj = h->pobj;
    if (d_exit_code) *d_exit_code = h->exit_code;
    return TMPC_OK;
}

int tmpc_pack_records(tmpc_handle *h, void *d_records, const void *d_guidance_id, const void *d_weight)
{
    if (!h || h->B <= 0 || !d_records) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_pack_records_kernel, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->B, h->pobj,
                       h->exit_code, (const int *)d_guidance_id, (const double *)d_weight, (tmpc_record *)d_records);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_select_best_records(tmpc_handle *h, const void *d_records, int32_t n_ranks, int32_t n_scenes, int32_t per_rank, void *d_best)
{
    if (!h || !d_records || !d_best || n_ranks <= 0 || n_scenes <= 0 || per_rank <= 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_select_best_records_kernel, dim3(n_scenes), dim3(64), 0, h->stream,
                       (const tmpc_record *)d_records, n_ranks, n_scenes, per_rank, (int *)d_best);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_gather_best(tmpc_handle *h, const void *d_best, int32_t n_sets, int32_t set_size, int32_t index_offset, void *d_xtraj, void *d_utraj)
{
    if (!h || !d_best || !d_xtraj || !d_utraj || n_sets <= 0 || set_size <= 0 || index_offset < 0 || (int64_t)n_sets * set_size > h->B) {
        if (h) h->err = "tmpc_gather_best: bad argument (n_sets x set_size entries of the current batch)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_gather_best_kernel, dim3(n_sets), dim3(64), 0, h->stream, (const int *)d_best, set_size, index_offset,
                       (h->d.N + 1) * tmpc::ext_nx(h->d), h->d.N * tmpc::NU, h->xtraj, h->utraj, (double *)d_xtraj, (double *)d_utraj);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_linearize_topology_ex(tmpc_handle *h, const void *d_obstacle_pos, int32_t n_obstacles, const void *d_obstacle_radius,
                               const void *d_static_halfspaces, int32_t n_static, const void *d_scene_of, const void *d_state_x,
                               double robot_radius, const void *d_is_original)
{
#ifdef TMPC_GENERATED_STAGE
    if (h) h->err = "tmpc_linearize_topology: not available in a generated solver (its parameter layout is the module stack's)";
    return TMPC_ERR_INVALID;
#endif
    if (!h || h->B <= 0 || !h->params || !d_scene_of || !d_state_x || h->d.n_lin <= 0 || n_obstacles < 0 || n_static < 0 ||
        n_obstacles + n_static > h->d.n_lin || (n_obstacles > 0 && !d_obstacle_pos) || (n_static > 0 && !d_static_halfspaces)) {
        if (h) h->err = "tmpc_linearize_topology: bad argument / no batch / more obstacle + static rows than the problem's topology rows";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * h->d.N;
    hipLaunchKernelGGL(tmpc::tmpc_linearize_topology_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B,
                       h->x0, const_cast<double *>(h->params), (const double *)d_obstacle_pos, (const int *)d_scene_of,
                       (const double *)d_state_x, robot_radius, (const uint8_t *)d_is_original, n_obstacles, (const double *)d_obstacle_radius,
                       (const double *)d_static_halfspaces, n_static);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_linearize_topology(tmpc_handle *h, const void *d_obstacle_pos, const void *d_scene_of, const void *d_state_x,
                            double robot_radius, const void *d_is_original)
{
    if (!h || !d_obstacle_pos) { if (h) h->err = "tmpc_linearize_topology: bad argument"; return TMPC_ERR_INVALID; }
    return tmpc_linearize_topology_ex(h, d_obstacle_pos, h->d.n_lin, nullptr, nullptr, 0, d_scene_of, d_state_x, robot_radius, d_is_original);
}

int tmpc_scenario_halfspaces(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_rows, const void *d_scene_of,
                             const void *d_state_x, double radius, double disc_offset)
{
#ifdef TMPC_GENERATED_STAGE
    if (h) h->err = "tmpc_scenario_halfspaces: not available in a generated solver (its parameter layout is the module stack's)";
    return TMPC_ERR_INVALID;
#endif
    if (!h || h->B <= 0 || !h->params || !d_samples || !d_scene_of || !d_state_x || n_pts <= 0 || n_rows <= 0 || n_rows > 64 ||
        n_rows > h->d.n_slk) {
        if (h) h->err = "tmpc_scenario_halfspaces: bad argument / no batch / more rows than the problem's slack rows";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t per_entry = 3 * sizeof(double) + sizeof(int);                        // a candidate: normal, margin, index word
    hipFuncAttributes fa;
    TMPC_HIP_CHECK(h, hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(tmpc::tmpc_scenario_halfspaces_kernel)));
    if ((size_t)n_pts * per_entry + fa.sharedSizeBytes > 160 * 1024) {      // the second pass's list (room for every sample) + the kernel's static tables
        h->err = "tmpc_scenario_halfspaces: that many samples per stage do not fit a workgroup's LDS (160 KiB minus the kernel's static tables: about 5480)";
        return TMPC_ERR_INVALID;
    }
    const size_t units = (size_t)h->B * h->d.N;
    const size_t need = units * n_rows + 1 + units + (size_t)h->B;  // rows' samples, the first pass's overflow list (count, units), empty-polygon stages per trajectory
    if (need > h->scn_cap) {
        if (h->scn_sample) { TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->scn_sample); h->scn_sample = nullptr; h->scn_cap = 0; }
        TMPC_HIP_CHECK(h, hipMalloc(&h->scn_sample, need * sizeof(int)));
        h->scn_cap = need;
    }
    h->scn_rows = n_rows; h->scn_B = h->B;
    // scenarios discarded for this batch (tmpc_scenario_discard after the batch was set) are left out of the polygons
    const bool use_discard = h->scn_discard && h->scn_discard_B == h->B && h->scn_discard_S > 0 && n_pts % h->scn_discard_S == 0;
    // first pass with a short candidate list (more workgroups per CU); second pass, with room for every sample, only for the
    // units the first pass recorded as not fitting
    int *overflow = h->scn_sample + units * n_rows;
    int *empty_stages = overflow + 1 + units;
    TMPC_HIP_CHECK(h, hipMemsetAsync(overflow, 0, sizeof(int), h->stream));
    TMPC_HIP_CHECK(h, hipMemsetAsync(empty_stages, 0, sizeof(int) * (size_t)h->B, h->stream));
    const int cap1 = n_pts < tmpc::POLY_LIST_CAP ? n_pts : tmpc::POLY_LIST_CAP;
    for (int pass = 0; pass < (cap1 < n_pts ? 2 : 1); pass++) {
        const int cap = pass == 0 ? cap1 : n_pts;
        const size_t lds = (size_t)cap * per_entry;
        if (lds > 48 * 1024)
            TMPC_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(tmpc::tmpc_scenario_halfspaces_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tmpc::tmpc_scenario_halfspaces_kernel, dim3(pass == 0 ? units : (units < 512 ? units : 512)), dim3(256), lds, h->stream, h->d, h->B, h->x0,
                           const_cast<double *>(h->params), (const double *)d_samples, n_pts, n_rows, (const int *)d_scene_of,
                           (const double *)d_state_x, radius, disc_offset, h->scn_sample, cap, overflow, pass, empty_stages,
                           use_discard ? h->scn_discard : nullptr, use_discard ? h->scn_discard_S : 1);
    }
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_sample_scenarios(tmpc_handle *h, const void *d_pred, const void *d_prob, int32_t n_solvers, int32_t n_obstacles, int32_t n_modes,
                          int32_t n_scenarios, uint64_t seed, void *d_samples)
{
    if (!h || !d_pred || !d_prob || !d_samples || n_solvers <= 0 || n_obstacles <= 0 || n_modes <= 0 || n_scenarios <= 0) {
        if (h) h->err = "tmpc_sample_scenarios: bad argument";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = n_solvers * n_obstacles * n_scenarios;
    hipLaunchKernelGGL(tmpc::tmpc_sample_scenarios_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.N, n_solvers, n_obstacles, n_modes,
                       n_scenarios, (unsigned long long)seed, (const double *)d_pred, (const double *)d_prob, (double *)d_samples);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_scenario_discard(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_scenarios, int32_t n_discard, const void *d_scene_of, double radius)
{
    if (!h || h->B <= 0 || !h->x0 || !d_samples || !d_scene_of || n_pts <= 0 || n_scenarios <= 0 || n_pts % n_scenarios != 0 || n_discard < 0 ||
        n_discard >= n_scenarios || n_scenarios > 16384) {
        if (h) h->err = "tmpc_scenario_discard: bad argument / no batch (n_pts = obstacles x n_scenarios, 0 <= n_discard < n_scenarios <= 16384)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t need = (size_t)h->B_max * n_scenarios;
    if (need > h->scn_discard_cap) {
        if (h->scn_discard) { TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->scn_discard); h->scn_discard = nullptr; h->scn_discard_cap = 0; }
        TMPC_HIP_CHECK(h, hipMalloc(&h->scn_discard, need));
        h->scn_discard_cap = need;
    }
    const size_t lds = (size_t)n_scenarios * sizeof(double);
    if (lds > 48 * 1024)
        TMPC_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(tmpc::tmpc_scenario_discard_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tmpc::tmpc_scenario_discard_kernel, dim3(h->B), dim3(256), lds, h->stream, h->d, h->B, h->x0, (const double *)d_samples, n_pts,
                       n_scenarios, (const int *)d_scene_of, radius, n_discard, h->scn_discard);
    TMPC_HIP_CHECK(h, hipGetLastError());
    h->scn_discard_S = n_scenarios; h->scn_discard_B = h->B; h->scn_discard_n = n_discard;
    return TMPC_OK;
}

int tmpc_scenario_discarded(tmpc_handle *h, void *d_mask)
{
    if (!h || !d_mask || !h->scn_discard || h->scn_discard_B != h->B) { if (h) h->err = "tmpc_scenario_discarded: no discard set for the current batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipMemcpyAsync(d_mask, h->scn_discard, (size_t)h->B * h->scn_discard_S, hipMemcpyDeviceToDevice, h->stream));
    return TMPC_OK;
}

int tmpc_scenario_empty_stages(tmpc_handle *h, void *d_count)
{
    if (!h || !d_count) return TMPC_ERR_INVALID;
    if (!h->scn_sample || h->scn_B != h->B || h->B <= 0) { h->err = "tmpc_scenario_empty_stages: the scenario rows of the current batch were not built by tmpc_scenario_halfspaces"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t units = (size_t)h->B * h->d.N;
    TMPC_HIP_CHECK(h, hipMemcpyAsync(d_count, h->scn_sample + units * h->scn_rows + 1 + units, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToDevice, h->stream));
    return TMPC_OK;
}

int tmpc_scenario_support(tmpc_handle *h, int32_t n_scenarios, double tol, void *d_support, void *d_active_rows)
{
    if (!h || !d_support || n_scenarios <= 0 || n_scenarios > 8192 || !(tol >= 0.0)) {
        if (h) h->err = "tmpc_scenario_support: bad argument (1 <= n_scenarios <= 8192, tol >= 0)";
        return TMPC_ERR_INVALID;
    }
    if (!h->scn_sample || h->scn_B != h->B || h->B <= 0 || !h->params) {
        h->err = "tmpc_scenario_support: the scenario rows of the current batch were not built by tmpc_scenario_halfspaces (call it after tmpc_set_batch)";
        return TMPC_ERR_INVALID;
    }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(tmpc::tmpc_scenario_support_kernel, dim3(h->B), dim3(64), 0, h->stream, h->d, h->B, h->params, h->xtraj,
                       h->scn_sample, h->scn_rows, n_scenarios, tol, (int *)d_support, (int *)d_active_rows);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_warmstart(tmpc_handle *h, const void *d_state, const void *d_mode, const void *d_src, double deceleration)
{
    if (!h || h->B <= 0 || !h->x0 || !h->xinit || !d_state) { if (h) h->err = "tmpc_warmstart: bad argument / no batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * (h->d.N + 1);
    hipLaunchKernelGGL(tmpc::tmpc_warmstart_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B, (const double *)d_state,
                       (const int *)d_mode, (const int *)d_src, h->xtraj, h->utraj, const_cast<double *>(h->x0),
                       const_cast<double *>(h->xinit), deceleration);
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_init_with_guidance(tmpc_handle *h, const void *d_gpos, const void *d_gvel, const void *d_enabled)
{
    if (!h || h->B <= 0 || !h->x0 || !d_gpos || !d_gvel) { if (h) h->err = "tmpc_init_with_guidance: bad argument / no batch"; return TMPC_ERR_INVALID; }
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int n = h->B * (h->d.N + 1);
    hipLaunchKernelGGL(tmpc::tmpc_init_with_guidance_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, h->B,
                       (const double *)d_gpos, (const double *)d_gvel, (const uint8_t *)d_enabled, const_cast<double *>(h->x0));
    TMPC_HIP_CHECK(h, hipGetLastError());
    return TMPC_OK;
}

int tmpc_debug_get_x0(tmpc_handle *h, double *x0, double *xinit)
{
    if (!h || h->B <= 0 || !h->x0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (x0) TMPC_HIP_CHECK(h, hipMemcpy(x0, h->x0, (size_t)h->B * (h->d.N + 1) * tmpc::ext_nv(h->d) * 8, hipMemcpyDeviceToHost));
    if (xinit) TMPC_HIP_CHECK(h, hipMemcpy(xinit, h->xinit, (size_t)h->B * tmpc::ext_nx(h->d) * 8, hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_enable_timing(tmpc_handle *h, int32_t max_records)
{
    if (!h || max_records < 0) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear(); h->ev_used = 0; h->timing = max_records > 0;
    h->ev.resize(2 * (size_t)max_records);
    for (auto &e : h->ev) TMPC_HIP_CHECK(h, hipEventCreate(&e));
    return TMPC_OK;
}

int tmpc_get_timings(tmpc_handle *h, float *ms, int32_t capacity, int32_t *n_out)
{
    if (!h || !ms || !n_out) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    int n = h->ev_used / 2; if (n > capacity) n = capacity;
    for (int i = 0; i < n; i++) TMPC_HIP_CHECK(h, hipEventElapsedTime(&ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
    *n_out = n; h->ev_used = 0;
    return TMPC_OK;
}

int tmpc_time_solve(tmpc_handle *h, int32_t reps, float *ms_each)
{
    if (!h || reps <= 0 || !ms_each) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    Events evs; evs.ev.assign(2 * (size_t)reps, nullptr);
    std::vector<hipEvent_t> &ev = evs.ev;
    for (auto &e : ev) TMPC_HIP_CHECK(h, hipEventCreate(&e));
    for (int i = 0; i < reps; i++) {
        TMPC_HIP_CHECK(h, hipEventRecord(ev[2 * i], h->stream));
        int rc = tmpc_solve(h);
        if (rc != TMPC_OK) return rc;
        TMPC_HIP_CHECK(h, hipEventRecord(ev[2 * i + 1], h->stream));
    }
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < reps; i++) TMPC_HIP_CHECK(h, hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]));
    return TMPC_OK;
}

int tmpc_debug_get_params(tmpc_handle *h, double *params)
{
    if (!h || h->B <= 0 || !params || !h->params) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    TMPC_HIP_CHECK(h, hipMemcpy(params, h->params, (size_t)h->B * h->d.N * h->d.npar * 8, hipMemcpyDeviceToHost));
    return TMPC_OK;
}

int tmpc_debug_profile(tmpc_handle *h, int64_t *cycles, int32_t n_phases)
{
    if (!h || h->B <= 0 || !cycles || n_phases < tmpc::PH_COUNT) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    DevBufs bufs;
    double *dp_ = nullptr;
    const size_t n = (size_t)h->B * tmpc::PH_COUNT;
    TMPC_HIP_CHECK(h, bufs.alloc(&dp_, n * 8));
    long long *dp = (long long *)dp_;
    TMPC_HIP_CHECK(h, hipMemset(dp, 0, n * 8));
    tmpc::SolveKernel pk = h->kernel;                   // the generic kernel profiles itself; fast shapes have an instrumented twin
    int thr = h->threads;
    size_t lds = h->fast ? h->lds_bytes_fast : h->lds_bytes;
    if (h->fast && h->threads == 128) lds = h->lds_bytes_fast2;
    if (h->fast && h->d.cost_model != 0) { h->err = "tmpc_debug_profile: no profiled twin for the curvature-aware cost"; return TMPC_ERR_INVALID; }
    if (h->fast) {
        pk = tmpc::pick_fast_kernel(h->d, &thr, true);
#ifndef TMPC_GENERATED_STAGE
        // the latency variants of cfg 2 are profiled as themselves (tmpc_set_latency_mode before the call)
        if (h->latency_mode == 2 && h->kernel_scan && h->scan_threads == 128 && h->scan_sl == 3 && h->d.n_up == 8 && h->d.M == 8) {
            pk = (tmpc::SolveKernel)tmpc::tmpc_solve_fast_kernel<8, 8, 6, 128, true, tmpc::ScanSolo>; thr = 128; lds = h->lds_bytes_scan;
        } else if (h->latency_mode != 0 && h->kernel_lat) {
            pk = tmpc::pick_latency_kernel(h->d, true); thr = 128; lds = h->lds_bytes_fast2;
        }
#endif
        TMPC_HIP_CHECK(h, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // (a compact handle is profiled through the fast kernel of its shape: same phases and arithmetic, one wave per SIMD)
    hipLaunchKernelGGL(pk, dim3(h->B), dim3(thr), lds, h->stream, h->d, h->B,
                       h->xinit, h->x0, h->params, h->xtraj, h->utraj, h->pobj, h->exit_code, h->qp_status,
                       h->sqp_iter, h->res_eq, h->qp_iter, dp, tmpc::StateIO{nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr});
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::vector<long long> host(n);
    TMPC_HIP_CHECK(h, hipMemcpy(host.data(), dp, n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < tmpc::PH_COUNT; i++) {
        double acc = 0.0;
        for (int b = 0; b < h->B; b++) acc += (double)host[(size_t)b * tmpc::PH_COUNT + i];
        cycles[i] = (int64_t)(acc / h->B);
    }
    return TMPC_OK;
}

int tmpc_debug_eval_stage(tmpc_handle *h, int32_t n, const double *z, const double *p, const double *pi, const double *lamh,
                          double *cost, double *cost_grad, double *cost_hess, double *hval, double *h_jac,
                          double *x_next, double *x_jac, double *lag_hess, double *mirror)
{
    if (!h || n <= 0 || !z || !p) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    const int nh = h->d.n_up + h->d.M;
    const size_t sz_in[4] = {(size_t)n * tmpc::ext_nv(h->d) * 8, (size_t)n * h->d.npar * 8, (size_t)n * 5 * 8, (size_t)n * nh * 8};
    const void *src[4] = {z, p, pi, lamh};
    DevBufs bufs;
    double *din[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
        if (!src[i]) continue;
        TMPC_HIP_CHECK(h, bufs.alloc(&din[i], sz_in[i]));
        TMPC_HIP_CHECK(h, hipMemcpy(din[i], src[i], sz_in[i], hipMemcpyHostToDevice));
    }
    const size_t sz_out[9] = {(size_t)n * 8, (size_t)n * 7 * 8, (size_t)n * 49 * 8, (size_t)n * nh * 8, (size_t)n * nh * 7 * 8,
                              (size_t)n * 5 * 8, (size_t)n * 35 * 8, (size_t)n * 49 * 8, (size_t)n * 49 * 8};
    double *dout[9]; void *dst[9] = {cost, cost_grad, cost_hess, hval, h_jac, x_next, x_jac, lag_hess, mirror};
    for (int i = 0; i < 9; i++) TMPC_HIP_CHECK(h, bufs.alloc(&dout[i], sz_out[i]));
    hipLaunchKernelGGL(tmpc::tmpc_debug_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d, n, din[0], din[1], din[2], din[3],
                       dout[0], dout[1], dout[2], dout[3], dout[4], dout[5], dout[6], dout[7], dout[8]);
    TMPC_HIP_CHECK(h, hipGetLastError());
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 9; i++)
        if (dst[i]) TMPC_HIP_CHECK(h, hipMemcpy(dst[i], dout[i], sz_out[i], hipMemcpyDeviceToHost));
    return TMPC_OK;
}

#ifdef TMPC_SWEEP_PROFILE
// Profiling build only (not declared in tmpc_hip.h): read and reset the sweep clock accumulators.
int tmpc_debug_sweep_profile(tmpc_handle *h, uint64_t *out, int32_t n)
{
    if (!h || !out || n < tmpc::SP_COUNT) return TMPC_ERR_INVALID;
    TMPC_HIP_CHECK(h, hipSetDevice(h->device));
    TMPC_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    unsigned long long host[tmpc::SP_COUNT];
    TMPC_HIP_CHECK(h, hipMemcpyFromSymbol(host, HIP_SYMBOL(tmpc::g_sweep_prof), sizeof host));
    for (int i = 0; i < tmpc::SP_COUNT; i++) out[i] = host[i];
    memset(host, 0, sizeof host);
    TMPC_HIP_CHECK(h, hipMemcpyToSymbol(HIP_SYMBOL(tmpc::g_sweep_prof), host, sizeof host));
    return TMPC_OK;
}
#endif

}  // extern "C"
#endif  // TMPC_SINGLE_KERNEL
