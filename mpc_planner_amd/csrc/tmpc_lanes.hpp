// mpc_planner_amd/csrc/tmpc_lanes.hpp -- THROUGHPUT variant of Solver::solve(): one LANE per trajectory.
//
// The wave-per-trajectory kernels (tmpc_solve.hip / tmpc_fast.hpp) keep a trajectory's whole solver state in the LDS +
// registers of one wavefront; a CU then holds four trajectories and the sequential Riccati sweeps use 7 of 64 lanes
// (DESIGN.md section 5).  This variant turns the mapping around for large batches (many control ticks / scenario solvers
// per launch): a wavefront owns 64 trajectories, lane = trajectory, and every lane runs the *scalar* SQP_RTI program
//   linearise -> MIRROR -> Mehrotra predictor-corrector IPM with a square-root Riccati recursion -> full step
// (Solver::solve, mpc_planner_solver/src/acados_solver_interface.cpp:86-204) on its own trajectory:
// no cross-lane traffic, no LDS, every VALU lane does useful work.  The per-trajectory state (iterate, stage blocks
// [W g | B A b], interior-point rows, Riccati factors) lives in an HBM workspace laid out field-major with the lane as the
// fastest index, so every load / store of the wave is one contiguous 512-byte segment; the reference-layout parameter
// tensor [B][N][npar] is transposed into that layout once per solve through LDS (coalesced on both sides).
// The kernel is bound by HBM bandwidth (the state is streamed a few times per interior-point iteration), not by capacity.
//
// An interior-point iteration is organised as four sweeps over the stages, fused so that every row is visited three times:
//   A  k = N..0   apply the previous step; residuals; barrier Hessian; Riccati factorisation; predictor backward sweep
//   B  k = 0..N   predictor forward sweep; affine row steps -> step length, centring parameter, corrector right-hand sides
//   C  k = N..0   corrector backward sweep (no row access)
//   D  k = 0..N   corrector forward sweep; row steps -> step length
// The algorithm, its constants and its stopping rules are those of the other kernels and of oracle/qp_ipm.c.
//
// The per-lane program is plain scalar code (TMPC_HD): the device kernel runs it with lane = threadIdx.x; tests/cpu_twin
// compiles the same source for the host to debug it against the oracle without a GPU (test infrastructure only).
#pragma once
#include "tmpc_stage.hpp"

// Keeps the instruction scheduler from hoisting one phase's loads into the previous phase (register pressure): the per-stage
// latency budget of this bandwidth-bound kernel allows a handful of dependent round trips.
#if defined(__HIP_DEVICE_COMPILE__)
#define TMPC_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TMPC_PHASE_FENCE() ((void)0)
#endif

// Optional phase clocks (profiling build only, -DTMPC_LANES_PROF: tools/profile_lanes.py): shader-clock cycles per phase,
// accumulated per lane and written to the first doubles of the lane's xinit padding... kept out of production builds.
#if defined(TMPC_LANES_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define TMPC_PROF_DECL long long prof_t0_ = clock64();
#define TMPC_PROF_ADD(slot) { const long long t1_ = clock64(); prof[slot] += t1_ - prof_t0_; prof_t0_ = t1_; }
#else
#define TMPC_PROF_DECL
#define TMPC_PROF_ADD(slot)
#endif

namespace tmpc {
namespace lanes {

constexpr int LW = 64;
enum { PF_LIN = 0, PF_A, PF_B, PF_C, PF_D, PF_STEP, PF_FINAL, PF_COUNT };            // trajectories per block = lanes of one wavefront

// Per-stage record of one lane (offsets in doubles).  Node N uses z, pi, v, dv, dpi, p only.
struct Layout {
    int N, nh, n_up, npar;
    int o_z, o_pi, o_v, o_dv, o_dpi, o_dva, o_p, o_y;      // iterate, QP iterate, directions, Riccati vectors
    int o_W, o_g, o_d8, o_b;                                // QP data of the current linearisation
    int o_L;                                                // Riccati factor: Lxu[5][2], L10, L00, L11, Lxx (15 packed)
    int o_ghp, o_rb, o_S1, o_S2;                            // predictor rhs, dynamics residual, corrector rhs parts
    int o_rows;                                             // general rows: D0, D1, D2, beta, t, lam (6 each)
    int o_box;                                              // box rows: (t, lam) x 14
    int sd;                                                 // doubles per stage record
    int o_par, o_xinit;                                     // block-level regions (doubles per lane): parameters, xinit
    int lane_doubles;                                       // doubles per lane in total
};

__host__ __device__ inline Layout make_layout(const Dims &d)
{
    Layout L;
    L.N = d.N; L.nh = d.n_up + d.M; L.n_up = d.n_up; L.npar = d.npar;
    int o = 0;
    auto take = [&](int n) { const int p = o; o += n; return p; };
    L.o_z = take(NV); L.o_pi = take(NX); L.o_v = take(NV); L.o_dv = take(NV); L.o_dpi = take(NX); L.o_dva = take(NV);
    L.o_p = take(NX); L.o_y = take(NU);
    L.o_W = take(NP28); L.o_g = take(NV); L.o_d8 = take(8); L.o_b = take(NX);
    L.o_L = take(28);
    L.o_ghp = take(NV); L.o_rb = take(NX); L.o_S1 = take(NV); L.o_S2 = take(NV);
    L.o_rows = take(6 * L.nh);
    L.o_box = take(2 * 14);
    L.sd = o;
    L.o_par = (d.N + 1) * L.sd;
    L.o_xinit = L.o_par + d.N * d.npar;
    L.lane_doubles = L.o_xinit + 8;
    return L;
}
__host__ __device__ inline size_t block_doubles(const Layout &L) { return (size_t)L.lane_doubles * LW; }

enum { D8_XA = 0, D8_XW, D8_XP, D8_XV, D8_YA, D8_YW, D8_YP, D8_YV };     // same order as the wave kernels' dyn8

struct Result { double pobj, res_eq; int exit_code, qp_status, sqp_iter, qp_iter; };

// ---- [B A] products from its 8 non-constant entries (solver_model.py:207-214 under ERK4: tmpc_stage.hpp dyn_jacobian) ----
TMPC_HD void ba_mul(const double *e, double dt, double hdt2, const double *v, double *o)
{
    o[0] = e[D8_XA] * v[ZA] + e[D8_XW] * v[ZW] + v[ZX] + e[D8_XP] * v[ZPSI] + e[D8_XV] * v[ZV];
    o[1] = e[D8_YA] * v[ZA] + e[D8_YW] * v[ZW] + v[ZY] + e[D8_YP] * v[ZPSI] + e[D8_YV] * v[ZV];
    o[2] = dt * v[ZW] + v[ZPSI];
    o[3] = dt * v[ZA] + v[ZV];
    o[4] = hdt2 * v[ZA] + dt * v[ZV] + v[ZS];
}
TMPC_HD void bat_mul(const double *e, double dt, double hdt2, const double *p, double *o)      // [B A]^T p
{
    o[ZA] = e[D8_XA] * p[0] + e[D8_YA] * p[1] + dt * p[3] + hdt2 * p[4];
    o[ZW] = e[D8_XW] * p[0] + e[D8_YW] * p[1] + dt * p[2];
    o[ZX] = p[0];
    o[ZY] = p[1];
    o[ZPSI] = e[D8_XP] * p[0] + e[D8_YP] * p[1] + p[2];
    o[ZV] = e[D8_XV] * p[0] + e[D8_YV] * p[1] + p[3] + dt * p[4];
    o[ZS] = p[4];
}
// y = P r = Lp (Lp^T r), Lp packed lower triangular 5x5
TMPC_HD void apply_P(const double *Lp, const double *r, double *y)
{
    double tmp[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        double a = 0.0;
#pragma unroll
        for (int l = i; l < NX; l++) a += Lp[pidx(l, i)] * r[l];
        tmp[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NX; i++) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l <= i; l++) a += Lp[pidx(i, l)] * tmp[l];
        y[i] = a;
    }
}

// Box row q of a stage: q < 4 inputs (var q >> 1), q >= 4 states (var 2 + (q - 4) >> 1); even = lower (+1), odd = upper (-1).
TMPC_HD constexpr int box_var(int q) { return q < 4 ? (q >> 1) : (NU + ((q - 4) >> 1)); }

// =====================================================================================================================
// The per-lane program.
// =====================================================================================================================
struct Lane {
    const Dims &d;
    const Layout &L;
    double *blk;          // workspace of the lane's 64-trajectory block (wave-uniform: field addresses stay in scalar registers)
    unsigned lane;        // element f of this lane: blk[f * LW + lane]
    long long *prof;      // profiling build: PF_COUNT cycle accumulators of this lane (nullptr otherwise)

    // (uniform row pointer)[lane]: the row address is scalar arithmetic, the lane offset one 32-bit VGPR (global_load saddr form)
    TMPC_HD double *row(int f) const { return blk + (size_t)f * LW; }
    TMPC_HD double &at(double *r) const { return *(double *)((char *)r + (lane << 3)); }          // uniform base + zext(32-bit lane offset)
    TMPC_HD double &F(int k, int off) const { return at(row(k * L.sd + off)); }
    TMPC_HD const double *par(int k) const { return &at(row(L.o_par + k * L.npar)); }
    TMPC_HD double xinit(int i) const { return at(row(L.o_xinit + i)); }
    TMPC_HD double slack() const { return d.slack ? xinit(NX) : 0.0; }
    TMPC_HD double sgn_of(int r) const { return r < L.n_up ? -1.0 : 1.0; }

    // The parameter rows are read-only during a solve: `__restrict__` tells the compiler that the workspace stores of the row
    // sink cannot touch them, so the parameter loads of a stage are issued together instead of one round trip per row.
    template <typename LamH, typename Sink>
    TMPC_HD void lin_stage(const double *__restrict__ p, const double *z, double pix, double piy, LamH lamh, Sink sink,
                           double (*W)[NV], double *g, double *BA, double *xn, double sl) const
    {
        stage_linearise(d, z, p, LW, pix, piy, lamh, sink, W, g, BA, xn, sl, nullptr);
    }

    // ---- linearisation of all stages at the current iterate + start of the QP (cold start of the interior-point rows) ----
    TMPC_HD void linearise() const
    {
        const int N = d.N;
        const double sl = slack();
        for (int k = 0; k < N; k++) {
            double z[NV], W[NV][NV], g[NV], BA[NX * NV], xn[NX];
#pragma unroll
            for (int i = 0; i < NV; i++) z[i] = F(k, L.o_z + i);
            // QP start: v = 0 (dx_0 = xinit - x_0), pi = 0
            double vx = 0.0, vy = 0.0, vp = 0.0;
            double v0[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) v0[i] = 0.0;
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < NX; i++) v0[i] = xinit(i) - z[NU + i];
                vx = v0[0]; vy = v0[1]; vp = v0[2];
            }
            double znx[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) znx[i] = F(k + 1, L.o_z + NU + i);
            const double pix = F(k + 1, L.o_pi + 0), piy = F(k + 1, L.o_pi + 1);
            auto lamh = [&](int r) { return -sgn_of(r) * F(k, L.o_rows + 6 * r + 5); };      // (lam_upper - lam_lower) of the last QP
            auto sink = [&](int r, const RowOut &ro) {
                const double s = sgn_of(r);
                const double beta = (r < L.n_up ? 0.0 : 1.0) - ro.h;
                const int o = L.o_rows + 6 * r;
                F(k, o + 0) = ro.gx; F(k, o + 1) = ro.gy; F(k, o + 2) = ro.gp; F(k, o + 3) = beta;
                const double r0 = s * (ro.gx * vx + ro.gy * vy + ro.gp * vp - beta);
                const double t = r0 > d.thr0 ? r0 : d.thr0;
                F(k, o + 4) = t; F(k, o + 5) = d.mu0 / t;
            };
            lin_stage(par(k), z, pix, piy, lamh, sink, W, g, BA, xn, sl);
            F(k, L.o_v + 0) = 0.0; F(k, L.o_v + 1) = 0.0;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                F(k, L.o_v + NU + i) = v0[i];
                F(k + 1, L.o_pi + i) = 0.0;                                     // QP multipliers start at 0 (after their use above)
            }
#pragma unroll
            for (int i = 0; i < NV; i++) F(k, L.o_g + i) = g[i];
            F(k, L.o_d8 + D8_XA) = BA[0 * NV + ZA]; F(k, L.o_d8 + D8_XW) = BA[0 * NV + ZW];
            F(k, L.o_d8 + D8_XP) = BA[0 * NV + ZPSI]; F(k, L.o_d8 + D8_XV) = BA[0 * NV + ZV];
            F(k, L.o_d8 + D8_YA) = BA[1 * NV + ZA]; F(k, L.o_d8 + D8_YW) = BA[1 * NV + ZW];
            F(k, L.o_d8 + D8_YP) = BA[1 * NV + ZPSI]; F(k, L.o_d8 + D8_YV) = BA[1 * NV + ZV];
#pragma unroll
            for (int i = 0; i < NX; i++) F(k, L.o_b + i) = xn[i] - znx[i];
            // box rows: r0 = sgn (v_i - (bound - z_i)) with v_i = 0 (the boxed entries of v start at 0; x_0 is fixed, not boxed)
#pragma unroll
            for (int q = 0; q < 14; q++) {
                const int i = box_var(q);
                const double r0 = (q & 1) ? d.ub[i] - z[i] : z[i] - d.lb[i];
                const double t = r0 > d.thr0 ? r0 : d.thr0;
                const bool act = q < 4 || k >= 1;
                F(k, L.o_box + 2 * q) = t; F(k, L.o_box + 2 * q + 1) = act ? d.mu0 / t : 0.0;
            }
            mirror7(W, d.reg_eps);
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) F(k, L.o_W + pidx(i, j)) = W[i][j];
        }
#pragma unroll
        for (int i = 0; i < NV; i++) F(N, L.o_v + i) = 0.0;
    }

    // ---- one QP: Mehrotra predictor-corrector, square-root Riccati.  Returns status (0 ok, 2 max iter, 3 min step, 4 NaN) ----
    // Memory discipline of every stage visit: all loads first (they are independent, so the wave has dozens of 512-byte
    // requests in flight), arithmetic on registers, stores last -- a store in the middle would pin every later load behind
    // it (same workspace, may alias).  General rows travel in chunks of RC rows for the same reason.
    static constexpr int RC = 8;

    TMPC_HD int ipm(int *iters_out) const
    {
        const int N = d.N, nh = L.nh;
        const double dt = d.dt, hdt2 = d.hdt2;
        const double m_rows = (double)(N * nh + 4 * N + 10 * (N - 1));
        const double le = sqrt(d.reg_eps);                     // chol of P_N = eps I
        int status = 2, iters = 0;
        double alpha = 0.0, smu = 0.0;
        bool first = true;
        for (int it = 0;; it++) {
            TMPC_PROF_DECL
            // ================= sweep A: k = N..0 =================
            double res_g = 0.0, res_b = 0.0, res_d = 0.0, res_m = 0.0, mu_sum = 0.0;
            bool bad = false;
            {   // terminal node: W_N = eps I on the states, no cost gradient, no rows
                double vN[NX], piN[NX], dvN[NX], dpN[NX];
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    vN[i] = F(N, L.o_v + NU + i); piN[i] = F(N, L.o_pi + i);
                    dvN[i] = first ? 0.0 : F(N, L.o_dv + NU + i); dpN[i] = first ? 0.0 : F(N, L.o_dpi + i);
                }
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    if (!first) { vN[i] += alpha * dvN[i]; piN[i] += alpha * dpN[i]; }
                    const double rg = d.reg_eps * vN[i] - piN[i];
                    res_g = fmax(res_g, fabs(rg));
                    dvN[i] = rg;
                }
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    if (!first) { F(N, L.o_v + NU + i) = vN[i]; F(N, L.o_pi + i) = piN[i]; }
                    F(N, L.o_p + i) = dvN[i];                                  // p_N = gh_N = rg_N (no rows at the terminal node)
                }
            }
            for (int k = N - 1; k >= 0; k--) {
                // ---------------- loads ----------------
                double v[NV], pik[NX], d8[8], dv[NV], dva[NV], dpk[NX];
#pragma unroll
                for (int i = 0; i < NV; i++) v[i] = F(k, L.o_v + i);
#pragma unroll
                for (int i = 0; i < NX; i++) pik[i] = k >= 1 ? F(k, L.o_pi + i) : 0.0;
#pragma unroll
                for (int i = 0; i < 8; i++) d8[i] = F(k, L.o_d8 + i);
#pragma unroll
                for (int i = 0; i < NV; i++) { dv[i] = first ? 0.0 : F(k, L.o_dv + i); dva[i] = first ? 0.0 : F(k, L.o_dva + i); }
#pragma unroll
                for (int i = 0; i < NX; i++) dpk[i] = (first || k < 1) ? 0.0 : F(k, L.o_dpi + i);
                double vo[NV];
#pragma unroll
                for (int i = 0; i < NV; i++) { vo[i] = v[i]; if (!first) v[i] += alpha * dv[i]; }
                if (!first && k >= 1) {
#pragma unroll
                    for (int i = 0; i < NX; i++) pik[i] += alpha * dpk[i];
                }
                // ---- rows: apply the step of the previous iteration, then residual / barrier terms at the new point ----
                double Hxx = 0, Hxy = 0, Hyy = 0, Hxp = 0, Hyp = 0, Hpp = 0;
                double hb[NV], rgs[NV], ghs[NV];
#pragma unroll
                for (int i = 0; i < NV; i++) { hb[i] = 0.0; rgs[i] = 0.0; ghs[i] = 0.0; }
                auto visit = [&](double cvo, double cdva, double cdv, double cvn, double sb, double &t, double &lam, double &dd_out, double &w_out) {
                    // signed dots of the row with v_old, dv_aff, dv, v_new; sb = sgn * beta
                    if (!first) {
                        const double rdo = cvo - sb - t;
                        const double it_ = 1.0 / t;
                        const double q0 = lam * t;
                        const double dta = cdva + rdo;
                        const double dla = -(q0 + lam * dta) * it_;
                        const double q = q0 - smu + dta * dla;
                        const double dtr = cdv + rdo;
                        const double dl = -(q + lam * dtr) * it_;
                        t += alpha * dtr; lam += alpha * dl;
                    }
                    const double rd = cvn - sb - t;
                    const double comp = lam * t;
                    const double dd = lam / t;
                    res_d = fmax(res_d, fabs(rd)); res_m = fmax(res_m, comp); mu_sum += comp;
                    dd_out = dd; w_out = dd * rd;
                };
                for (int r0 = 0; r0 < nh; r0 += RC) {
                    double c0[RC], c1[RC], c2[RC], sb[RC], t[RC], lam[RC];
#pragma unroll
                    for (int j = 0; j < RC; j++) {
                        const int r = r0 + j < nh ? r0 + j : nh - 1;
                        const int o = L.o_rows + 6 * r;
                        const double s = sgn_of(r);
                        c0[j] = s * F(k, o + 0); c1[j] = s * F(k, o + 1); c2[j] = s * F(k, o + 2); sb[j] = s * F(k, o + 3);
                        t[j] = F(k, o + 4); lam[j] = F(k, o + 5);
                    }
#pragma unroll
                    for (int j = 0; j < RC; j++) {
                        if (r0 + j < nh) {
                            double dd, w;
                            visit(c0[j] * vo[ZX] + c1[j] * vo[ZY] + c2[j] * vo[ZPSI], c0[j] * dva[ZX] + c1[j] * dva[ZY] + c2[j] * dva[ZPSI],
                                  c0[j] * dv[ZX] + c1[j] * dv[ZY] + c2[j] * dv[ZPSI], c0[j] * v[ZX] + c1[j] * v[ZY] + c2[j] * v[ZPSI],
                                  sb[j], t[j], lam[j], dd, w);
                            rgs[ZX] += lam[j] * c0[j]; rgs[ZY] += lam[j] * c1[j]; rgs[ZPSI] += lam[j] * c2[j];
                            ghs[ZX] += w * c0[j]; ghs[ZY] += w * c1[j]; ghs[ZPSI] += w * c2[j];
                            const double d0 = dd * c0[j], d1 = dd * c1[j], d2 = dd * c2[j];
                            Hxx += d0 * c0[j]; Hxy += d1 * c0[j]; Hyy += d1 * c1[j]; Hxp += d2 * c0[j]; Hyp += d2 * c1[j]; Hpp += d2 * c2[j];
                        }
                    }
                    if (!first) {
#pragma unroll
                        for (int j = 0; j < RC; j++)
                            if (r0 + j < nh) { F(k, L.o_rows + 6 * (r0 + j) + 4) = t[j]; F(k, L.o_rows + 6 * (r0 + j) + 5) = lam[j]; }
                    }
                }
                // box rows; afterwards the stage's QP data and what node k+1 left behind (chol of P_{k+1}, p_{k+1}, dx_{k+1}, pi_{k+1}:
                // re-read rather than carried in registers across the rows)
                TMPC_PHASE_FENCE();
                double zk[NV], bt[14], bl[14];
#pragma unroll
                for (int i = 0; i < NV; i++) zk[i] = F(k, L.o_z + i);
#pragma unroll
                for (int q = 0; q < 14; q++) { bt[q] = F(k, L.o_box + 2 * q); bl[q] = F(k, L.o_box + 2 * q + 1); }
#pragma unroll
                for (int q = 0; q < 14; q++) {
                    if (q < 4 || k >= 1) {
                        const int i = box_var(q);
                        const double s = (q & 1) ? -1.0 : 1.0;
                        const double sbq = s * (((q & 1) ? d.ub[i] : d.lb[i]) - zk[i]);
                        double dd, w;
                        visit(s * vo[i], s * dva[i], s * dv[i], s * v[i], sbq, bt[q], bl[q], dd, w);
                        rgs[i] += bl[q] * s; ghs[i] += w * s; hb[i] += dd;
                    }
                }
                TMPC_PHASE_FENCE();
                double Wk[NP28], gk[NV], bk[NX], Lp[15], pn[NX], vnx[NX], pin[NX];
                load_Lp(k + 1, Lp);
#pragma unroll
                for (int i = 0; i < NX; i++) { pn[i] = F(k + 1, L.o_p + i); vnx[i] = F(k + 1, L.o_v + NU + i); pin[i] = F(k + 1, L.o_pi + i); }
#pragma unroll
                for (int e = 0; e < NP28; e++) Wk[e] = F(k, L.o_W + e);
#pragma unroll
                for (int i = 0; i < NV; i++) gk[i] = F(k, L.o_g + i);
#pragma unroll
                for (int i = 0; i < NX; i++) bk[i] = F(k, L.o_b + i);
                // ---- stage residuals: rg = g + W v + [B A]^T pi_{k+1} - [0; pi_k] - sum sgn lam c;  rb = b - dx_{k+1} + [B A] v ----
                double rg[NV], tmp7[NV], rb[NX];
                bat_mul(d8, dt, hdt2, pin, tmp7);
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    double acc = gk[i];
#pragma unroll
                    for (int j = 0; j < NV; j++) acc += Wk[sidx(i, j)] * v[j];
                    acc += tmp7[i];
                    if (i >= NU) acc -= pik[i - NU];
                    rg[i] = acc;
                }
                double gh[NV];
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    const double full = rg[i] - rgs[i];
                    const bool fixed = k == 0 && i >= NU;                     // dx_0 is fixed: no stationarity residual
                    if (!fixed) res_g = fmax(res_g, fabs(full));
                    gh[i] = fixed ? 0.0 : rg[i] + ghs[i];                      // predictor rhs: rg0 + sum (lam/t) rd c
                }
                ba_mul(d8, dt, hdt2, v, rb);
#pragma unroll
                for (int i = 0; i < NX; i++) { rb[i] += bk[i] - vnx[i]; res_b = fmax(res_b, fabs(rb[i])); }
                // ---- barrier-augmented Hessian + square-root Riccati step: F = Hh + G^T G, G = Lp^T [B A] ----
                Wk[pidx(ZX, ZX)] += Hxx; Wk[pidx(ZY, ZX)] += Hxy; Wk[pidx(ZY, ZY)] += Hyy;
                Wk[pidx(ZPSI, ZX)] += Hxp; Wk[pidx(ZPSI, ZY)] += Hyp; Wk[pidx(ZPSI, ZPSI)] += Hpp;
#pragma unroll
                for (int i = 0; i < NV; i++) Wk[pidx(i, i)] += hb[i];
                {
                    // G = Lp^T [B A] (5 x 7).  Columns x, y, spline of [B A] are unit vectors, so those columns of G are rows 0, 1, 4
                    // of Lp (G[l][x] = Lp[0][l], ...: zero below the diagonal structure); the other four are dense.
                    double Ga[NX], Gw[NX], Gp[NX], Gv[NX];
#pragma unroll
                    for (int i = 0; i < NX; i++) {
                        // (Lp^T)_{il} = Lp[l][i], l >= i; ascending l like the dense product
                        double ga = 0.0, gw = 0.0, gp = 0.0, gv = 0.0;
                        if (i <= 0) { ga += Lp[pidx(0, i)] * d8[D8_XA]; gw += Lp[pidx(0, i)] * d8[D8_XW]; gp += Lp[pidx(0, i)] * d8[D8_XP]; gv += Lp[pidx(0, i)] * d8[D8_XV]; }
                        if (i <= 1) { ga += Lp[pidx(1, i)] * d8[D8_YA]; gw += Lp[pidx(1, i)] * d8[D8_YW]; gp += Lp[pidx(1, i)] * d8[D8_YP]; gv += Lp[pidx(1, i)] * d8[D8_YV]; }
                        if (i <= 2) { gw += Lp[pidx(2, i)] * dt; gp += Lp[pidx(2, i)]; }
                        if (i <= 3) { ga += Lp[pidx(3, i)] * dt; gv += Lp[pidx(3, i)]; }
                        ga += Lp[pidx(4, i)] * hdt2; gv += Lp[pidx(4, i)] * dt;
                        Ga[i] = ga; Gw[i] = gw; Gp[i] = gp; Gv[i] = gv;
                    }
                    // entry (l, col) of G and whether it can be non-zero
                    auto nz = [](int l, int col) { return col == ZX ? l == 0 : (col == ZY ? l <= 1 : true); };
                    auto Gv_ = [&](int l, int col) {
                        return col == ZA ? Ga[l] : col == ZW ? Gw[l] : col == ZPSI ? Gp[l] : col == ZV ? Gv[l]
                             : col == ZX ? Lp[pidx(0, l <= 0 ? l : 0)] : col == ZY ? Lp[pidx(1, l <= 1 ? l : 0)] : Lp[pidx(4, l)];
                    };
#pragma unroll
                    for (int i = 0; i < NV; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) {
                            double acc = Wk[pidx(i, j)];
#pragma unroll
                            for (int l = 0; l < NX; l++)
                                if (nz(l, i) && nz(l, j)) acc += Gv_(l, i) * Gv_(l, j);
                            Wk[pidx(i, j)] = acc;
                        }
                }
                // Cholesky in place (lower, packed)
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    double dj = Wk[pidx(j, j)];
#pragma unroll
                    for (int l = 0; l < j; l++) dj -= Wk[pidx(j, l)] * Wk[pidx(j, l)];
                    if (!(dj > 0.0)) bad = true;
                    const double ljj = sqrt(dj);
                    Wk[pidx(j, j)] = ljj;
#pragma unroll
                    for (int i = j + 1; i < NV; i++) {
                        double a = Wk[pidx(i, j)];
#pragma unroll
                        for (int l = 0; l < j; l++) a -= Wk[pidx(i, l)] * Wk[pidx(j, l)];
                        Wk[pidx(i, j)] = a / ljj;
                    }
                }
                // ---- predictor backward sweep: f = gh + [B A]^T (P rb + p_{k+1});  y = Luu^-1 f_u;  p_k = f_x - Lxu y ----
                double Pb[NX], f[NV];
                apply_P(Lp, rb, Pb);
#pragma unroll
                for (int i = 0; i < NX; i++) Pb[i] += pn[i];
                bat_mul(d8, dt, hdt2, Pb, f);
#pragma unroll
                for (int i = 0; i < NV; i++) f[i] += gh[i];
                const double y0 = f[0] / Wk[pidx(0, 0)];
                const double y1 = (f[1] - Wk[pidx(1, 0)] * y0) / Wk[pidx(1, 1)];
#pragma unroll
                for (int i = 0; i < NX; i++) pn[i] = f[NU + i] - Wk[pidx(NU + i, 0)] * y0 - Wk[pidx(NU + i, 1)] * y1;
                // ---------------- stores ----------------
                if (!first) {
#pragma unroll
                    for (int i = 0; i < NV; i++) F(k, L.o_v + i) = v[i];
                    if (k >= 1) {
#pragma unroll
                        for (int i = 0; i < NX; i++) F(k, L.o_pi + i) = pik[i];
                    }
#pragma unroll
                    for (int q = 0; q < 14; q++)
                        if (q < 4 || k >= 1) { F(k, L.o_box + 2 * q) = bt[q]; F(k, L.o_box + 2 * q + 1) = bl[q]; }
                }
                F(k, L.o_y + 0) = y0; F(k, L.o_y + 1) = y1;
#pragma unroll
                for (int i = 0; i < NX; i++) F(k, L.o_p + i) = pn[i];
#pragma unroll
                for (int e = 0; e < NP28; e++) F(k, L.o_L + e) = Wk[e];
#pragma unroll
                for (int i = 0; i < NV; i++) F(k, L.o_ghp + i) = gh[i];
#pragma unroll
                for (int i = 0; i < NX; i++) F(k, L.o_rb + i) = rb[i];
            }
            TMPC_PROF_ADD(PF_A)
            const double mu = mu_sum / m_rows;
            if (!(__builtin_isfinite(res_g) && __builtin_isfinite(res_b) && __builtin_isfinite(res_d) && __builtin_isfinite(res_m))) { status = 4; break; }
            if (res_g <= d.qp_tol && res_b <= d.qp_tol && res_d <= d.qp_tol && res_m <= d.qp_tol) { status = 0; break; }
            if (it >= d.qp_iter_max) { status = 2; break; }
            iters = it + 1;
            if (bad) { status = 4; break; }

            // ================= sweep B: k = 0..N-1, predictor forward + affine row steps =================
            double gmax = 0.0, A1 = 0.0, A2 = 0.0;
            {
                double dx[NX];
#pragma unroll
                for (int i = 0; i < NX; i++) dx[i] = 0.0;
                for (int k = 0; k < N; k++) {
                    double dva[NV], v[NV], zk[NV], d8[8], rbk[NX], bt[14], bl[14], fw[15];
                    load_forward(k, fw, d8);
#pragma unroll
                    for (int i = 0; i < NV; i++) { v[i] = F(k, L.o_v + i); zk[i] = F(k, L.o_z + i); }
#pragma unroll
                    for (int i = 0; i < NX; i++) rbk[i] = F(k, L.o_rb + i);
#pragma unroll
                    for (int q = 0; q < 14; q++) { bt[q] = F(k, L.o_box + 2 * q); bl[q] = F(k, L.o_box + 2 * q + 1); }
                    forward_step(fw, dx, dva);
                    double S1[NV], S2[NV];
#pragma unroll
                    for (int i = 0; i < NV; i++) { S1[i] = 0.0; S2[i] = 0.0; }
                    auto visit = [&](double cv, double cdva, double sb, double t, double lam, double &s1, double &s2) {
                        const double rd = cv - sb - t;
                        const double it_ = 1.0 / t;
                        const double dta = cdva + rd;
                        const double dla = -(lam * t + lam * dta) * it_;
                        // step to the boundary: alpha_max = 1 / max(-dt/t, -dlam/lam), kept as the running maximum of the ratios
                        // (no division, no divergent branch per row); here -dlam/lam = (t + dt)/t
                        gmax = fmax(gmax, fmax(-dta * it_, (t + dta) * it_));
                        A1 += lam * dta + t * dla; A2 += dla * dta;
                        s1 = dta * dla * it_; s2 = it_;
                    };
                    for (int r0 = 0; r0 < nh; r0 += RC) {
                        double c0[RC], c1[RC], c2[RC], sb[RC], t[RC], lam[RC];
#pragma unroll
                        for (int j = 0; j < RC; j++) {
                            const int r = r0 + j < nh ? r0 + j : nh - 1;
                            const int o = L.o_rows + 6 * r;
                            const double s = sgn_of(r);
                            c0[j] = s * F(k, o + 0); c1[j] = s * F(k, o + 1); c2[j] = s * F(k, o + 2); sb[j] = s * F(k, o + 3);
                            t[j] = F(k, o + 4); lam[j] = F(k, o + 5);
                        }
#pragma unroll
                        for (int j = 0; j < RC; j++) {
                            if (r0 + j < nh) {
                                double s1, s2;
                                visit(c0[j] * v[ZX] + c1[j] * v[ZY] + c2[j] * v[ZPSI], c0[j] * dva[ZX] + c1[j] * dva[ZY] + c2[j] * dva[ZPSI],
                                      sb[j], t[j], lam[j], s1, s2);
                                S1[ZX] += s1 * c0[j]; S1[ZY] += s1 * c1[j]; S1[ZPSI] += s1 * c2[j];
                                S2[ZX] += s2 * c0[j]; S2[ZY] += s2 * c1[j]; S2[ZPSI] += s2 * c2[j];
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 14; q++) {
                        if (q < 4 || k >= 1) {
                            const int i = box_var(q);
                            const double s = (q & 1) ? -1.0 : 1.0;
                            const double sbq = s * (((q & 1) ? d.ub[i] : d.lb[i]) - zk[i]);
                            double s1, s2;
                            visit(s * v[i], s * dva[i], sbq, bt[q], bl[q], s1, s2);
                            S1[i] += s1 * s; S2[i] += s2 * s;
                        }
                    }
                    double dxn[NX];
                    ba_mul(d8, dt, hdt2, dva, dxn);
#pragma unroll
                    for (int i = 0; i < NX; i++) dx[i] = dxn[i] + rbk[i];
#pragma unroll
                    for (int i = 0; i < NV; i++) { F(k, L.o_dva + i) = dva[i]; F(k, L.o_S1 + i) = S1[i]; F(k, L.o_S2 + i) = S2[i]; }
                }
            }
            TMPC_PROF_ADD(PF_B)
            const double a_aff = gmax > 1.0 ? 1.0 / gmax : 1.0;
            const double mu_aff = (mu_sum + a_aff * A1 + a_aff * a_aff * A2) / m_rows;
            double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
            sigma = sigma * sigma * sigma;
            smu = sigma * mu;

            // ================= sweep C: k = N..0, corrector backward (rhs = predictor rhs + S1 - sigma mu S2) =================
            {
                double Lp[15], pn[NX];
#pragma unroll
                for (int i = 0; i < 15; i++) Lp[i] = 0.0;
#pragma unroll
                for (int i = 0; i < NX; i++) { Lp[pidx(i, i)] = le; pn[i] = F(N, L.o_p + i); }
                for (int k = N - 1; k >= 0; k--) {
                    double d8[8], rb[NX], ghp[NV], S1[NV], S2[NV], Lk[NP28], Pb[NX], f[NV];
#pragma unroll
                    for (int i = 0; i < 8; i++) d8[i] = F(k, L.o_d8 + i);
#pragma unroll
                    for (int i = 0; i < NX; i++) rb[i] = F(k, L.o_rb + i);
#pragma unroll
                    for (int i = 0; i < NV; i++) { ghp[i] = F(k, L.o_ghp + i); S1[i] = F(k, L.o_S1 + i); S2[i] = F(k, L.o_S2 + i); }
#pragma unroll
                    for (int e = 0; e < NP28; e++) Lk[e] = F(k, L.o_L + e);
                    apply_P(Lp, rb, Pb);
#pragma unroll
                    for (int i = 0; i < NX; i++) Pb[i] += pn[i];
                    bat_mul(d8, dt, hdt2, Pb, f);
#pragma unroll
                    for (int i = 0; i < NV; i++) {
                        const bool fixed = k == 0 && i >= NU;
                        f[i] += fixed ? 0.0 : ghp[i] + (S1[i] - smu * S2[i]);
                    }
                    const double y0 = f[0] / Lk[pidx(0, 0)];
                    const double y1 = (f[1] - Lk[pidx(1, 0)] * y0) / Lk[pidx(1, 1)];
#pragma unroll
                    for (int i = 0; i < NX; i++) pn[i] = f[NU + i] - Lk[pidx(NU + i, 0)] * y0 - Lk[pidx(NU + i, 1)] * y1;
#pragma unroll
                    for (int i = 0; i < NX; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) Lp[pidx(i, j)] = Lk[pidx(NU + i, NU + j)];
                    F(k, L.o_y + 0) = y0; F(k, L.o_y + 1) = y1;
#pragma unroll
                    for (int i = 0; i < NX; i++) F(k, L.o_p + i) = pn[i];
                }
            }

            TMPC_PROF_ADD(PF_C)
            // ================= sweep D: k = 0..N-1, corrector forward + row steps =================
            gmax = 0.0;
            {
                double dx[NX];
#pragma unroll
                for (int i = 0; i < NX; i++) dx[i] = 0.0;
                for (int k = 0; k < N; k++) {
                    double dv[NV], dva[NV], v[NV], zk[NV], d8[8], rbk[NX], bt[14], bl[14], fw[15], Lpn[15], pnx[NX];
                    load_forward(k, fw, d8);
#pragma unroll
                    for (int i = 0; i < NV; i++) { v[i] = F(k, L.o_v + i); zk[i] = F(k, L.o_z + i); dva[i] = F(k, L.o_dva + i); }
#pragma unroll
                    for (int i = 0; i < NX; i++) { rbk[i] = F(k, L.o_rb + i); pnx[i] = F(k + 1, L.o_p + i); }
#pragma unroll
                    for (int q = 0; q < 14; q++) { bt[q] = F(k, L.o_box + 2 * q); bl[q] = F(k, L.o_box + 2 * q + 1); }
                    load_Lp(k + 1, Lpn);
                    forward_step(fw, dx, dv);
                    auto visit = [&](double cv, double cdva, double cdv, double sb, double t, double lam) {
                        const double rd = cv - sb - t;
                        const double it_ = 1.0 / t;
                        const double q0 = lam * t;
                        const double dta = cdva + rd;
                        const double dla = -(q0 + lam * dta) * it_;
                        const double q = q0 - smu + dta * dla;
                        const double dtr = cdv + rd;
                        const double dl = -(q + lam * dtr) * it_;
                        gmax = fmax(gmax, fmax(-dtr * it_, -dl / lam));
                    };
                    for (int r0 = 0; r0 < nh; r0 += RC) {
                        double c0[RC], c1[RC], c2[RC], sb[RC], t[RC], lam[RC];
#pragma unroll
                        for (int j = 0; j < RC; j++) {
                            const int r = r0 + j < nh ? r0 + j : nh - 1;
                            const int o = L.o_rows + 6 * r;
                            const double s = sgn_of(r);
                            c0[j] = s * F(k, o + 0); c1[j] = s * F(k, o + 1); c2[j] = s * F(k, o + 2); sb[j] = s * F(k, o + 3);
                            t[j] = F(k, o + 4); lam[j] = F(k, o + 5);
                        }
#pragma unroll
                        for (int j = 0; j < RC; j++)
                            if (r0 + j < nh)
                                visit(c0[j] * v[ZX] + c1[j] * v[ZY] + c2[j] * v[ZPSI], c0[j] * dva[ZX] + c1[j] * dva[ZY] + c2[j] * dva[ZPSI],
                                      c0[j] * dv[ZX] + c1[j] * dv[ZY] + c2[j] * dv[ZPSI], sb[j], t[j], lam[j]);
                    }
#pragma unroll
                    for (int q = 0; q < 14; q++) {
                        if (q < 4 || k >= 1) {
                            const int i = box_var(q);
                            const double s = (q & 1) ? -1.0 : 1.0;
                            const double sbq = s * (((q & 1) ? d.ub[i] : d.lb[i]) - zk[i]);
                            visit(s * v[i], s * dva[i], s * dv[i], sbq, bt[q], bl[q]);
                        }
                    }
                    double dxn[NX], dpi[NX];
                    ba_mul(d8, dt, hdt2, dv, dxn);
#pragma unroll
                    for (int i = 0; i < NX; i++) dx[i] = dxn[i] + rbk[i];
                    apply_P(Lpn, dx, dpi);                                   // dpi_{k+1} = P_{k+1} dx_{k+1} + p_{k+1}
#pragma unroll
                    for (int i = 0; i < NV; i++) F(k, L.o_dv + i) = dv[i];
#pragma unroll
                    for (int i = 0; i < NX; i++) F(k + 1, L.o_dpi + i) = dpi[i] + pnx[i];
                }
#pragma unroll
                for (int i = 0; i < NX; i++) F(N, L.o_dv + NU + i) = dx[i];
            }
            TMPC_PROF_ADD(PF_D)
            alpha = 0.999 > gmax ? 1.0 : 0.999 / gmax;                     // min(1, 0.999 alpha_max)
            if (!__builtin_isfinite(gmax)) { status = 4; break; }
            if (alpha < 1e-12) { status = 3; break; }
            first = false;
        }
        *iters_out = iters;
        return status;
    }

    // chol of P_k: Lxx of stage k's factor, or sqrt(eps) I at the terminal node
    TMPC_HD void load_Lp(int k, double *Lp) const
    {
        if (k >= d.N) {
            const double le = sqrt(d.reg_eps);
#pragma unroll
            for (int i = 0; i < 15; i++) Lp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < NX; i++) Lp[pidx(i, i)] = le;
        } else {
#pragma unroll
            for (int i = 0; i < NX; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) Lp[pidx(i, j)] = F(k, L.o_L + pidx(NU + i, NU + j));
        }
    }
    // operands of one forward step: fw = [y0, y1, l00, l10, l11, Lxu (5 x 2)], and the stage's dyn8
    TMPC_HD void load_forward(int k, double *fw, double *d8) const
    {
#pragma unroll
        for (int i = 0; i < 8; i++) d8[i] = F(k, L.o_d8 + i);
        fw[0] = F(k, L.o_y + 0); fw[1] = F(k, L.o_y + 1);
        fw[2] = F(k, L.o_L + pidx(0, 0)); fw[3] = F(k, L.o_L + pidx(1, 0)); fw[4] = F(k, L.o_L + pidx(1, 1));
#pragma unroll
        for (int j = 0; j < NX; j++) { fw[5 + 2 * j] = F(k, L.o_L + pidx(NU + j, 0)); fw[6 + 2 * j] = F(k, L.o_L + pidx(NU + j, 1)); }
    }
    // du = -Luu^-T (Lxu^T dx + y);  dvec = [du; dx]
    TMPC_HD void forward_step(const double *fw, const double *dx, double *dvec) const
    {
        double r0 = fw[0], r1 = fw[1];
#pragma unroll
        for (int j = 0; j < NX; j++) { r0 += fw[5 + 2 * j] * dx[j]; r1 += fw[6 + 2 * j] * dx[j]; }
        const double u1 = -r1 / fw[4];
        const double u0 = (-r0 - fw[3] * u1) / fw[2];
        dvec[0] = u0; dvec[1] = u1;
#pragma unroll
        for (int j = 0; j < NX; j++) dvec[NU + j] = dx[j];
    }

    // ---- tmpc_solve_iterations protocol: a slot whose QP stopped with qp_status != 0 has left the reference's loop (:105-106);
    // a solve that did not succeed resets the capsule's multipliers (:187-191) ----
    TMPC_HD bool stopped() const { return at(row(L.o_xinit + 7)) != 0.0; }
    TMPC_HD void close_call(const Result &R, bool complete) const
    {
        if (R.sqp_iter > 0) at(row(L.o_xinit + 7)) = R.qp_status != 0 ? 1.0 : 0.0;
        if (complete && R.exit_code != 1) {
            for (int k = 0; k <= d.N; k++) {
#pragma unroll
                for (int i = 0; i < NX; i++) F(k, L.o_pi + i) = 0.0;
                if (k < d.N)
                    for (int r = 0; r < L.nh; r++) F(k, L.o_rows + 6 * r + 5) = 0.0;
            }
        }
    }

    // ---- n_iter RTI iterations from the lane's current (z, pi, lam) + completeOneIteration (:162-204) ----
    TMPC_HD Result solve(int n_iter) const
    {
        const int N = d.N;
        Result R;
        int status = 0;
        R.qp_status = 0; R.sqp_iter = 0; R.qp_iter = 0;
        for (int it = 0; it < n_iter; it++) {
            TMPC_PROF_DECL
            linearise();
            TMPC_PROF_ADD(PF_LIN)
            int iters = 0;
            R.qp_status = ipm(&iters);
            R.sqp_iter = it + 1; R.qp_iter += iters;
            if (R.qp_status != 0 && R.qp_status != 2) { status = 4; break; }       // ACADOS_QP_FAILURE, no step (DESIGN U5)
            status = 0;
            for (int k = 0; k <= N; k++) {                                          // full step; pi, lam are the QP's already
#pragma unroll
                for (int i = 0; i < NV; i++)
                    if (!(k == N && i < NU)) F(k, L.o_z + i) += F(k, L.o_v + i);
            }
            if (R.qp_status != 0) break;
        }
        // cost, defects, initial-condition violation
        double cost = 0.0, res = 0.0;
        const double sl = slack();
        for (int k = 0; k < N; k++) {
            double z[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) z[i] = F(k, L.o_z + i);
            CostOut co;
            cost_eval(d, z, par(k), LW, co, false, sl);
            DynOut dy;
            dyn_eval(d, z, dy, false);
            cost += d.dt * co.val;
#pragma unroll
            for (int i = 0; i < NX; i++) res = fmax(res, fabs(dy.xn[i] - F(k + 1, L.o_z + NU + i)));
        }
#pragma unroll
        for (int i = 0; i < NX; i++) res = fmax(res, fabs(F(0, L.o_z + NU + i) - xinit(i)));
        if (res > 1e-2 && status == 0) status = 4;
        if (!__builtin_isfinite(cost)) status = 4;
        R.pobj = cost; R.res_eq = res;
        R.exit_code = status == 0 ? 1 : (status == 1 ? 0 : status);                 // Forces-style mapping (:197-201)
        return R;
    }
};

}  // namespace lanes
}  // namespace tmpc
