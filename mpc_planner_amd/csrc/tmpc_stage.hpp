// mpc_planner_amd/csrc/tmpc_stage.hpp -- gfx950 device functions: NLP stage functions with hand-derived
// first/second derivatives, and MIRROR regularisation.  HIP only (no CUDA-compat paths).
//
// What the reference gets from CasADi codegen + acados modules at solver-generation time
// (solver_generator/generate_acados_solver.py:27-65,190) is written out analytically here:
//   dynamics   solver_generator/solver_model.py:207-214 under acados ERK4 x 3 (generate_acados_solver.py:148-150)
//   cost       mpc_planner_modules/scripts/mpc_base.py:47-60, contouring.py:48-98, solver_generator/spline.py:28-77
//   rows h     mpc_planner_modules/scripts/guidance_constraints.py:95-110, ellipsoid_constraints.py:65-119
//   MIRROR     regularize_method (generate_acados_solver.py:157)
//   rows h     ... decomp_constraints.py:68-98, scenario_constraints.py:64-94 (halfspaces relaxed by the slack state)
// z = [a, w, x, y, psi, v, spline] (inputs first: solver_model.py:118-128).
//
// Slack model (ContouringSecondOrderUnicycleModelWithSlack, solver_model.py:274-298: one more state with slack' = 0).
// acados fixes every state at node 0 (ocp.constraints.x0, generate_acados_solver.py:95) and the slack state has no
// dynamics, so each QP pins it: after any full step slack_k = xinit_slack at every node, whatever the warm start held.
// The kernels therefore carry the 7 variables above and treat slack as the per-trajectory constant xinit[5]: it
// relaxes the decomp / scenario rows, adds dt * w_slack * slack^2 per stage to the objective, and is written back into
// the 6th state column of the outputs.  (Exact: the QP is strictly convex after MIRROR, so substituting the pinned
// chain does not change its solution; the CPU oracle does the same substitution and tests/test_oracle_solve_slack.py
// checks it against the full 8-variable NLP.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Generated solvers (mpc_planner_amd/codegen): -DTMPC_GENERATED_STAGE='"<header>"' replaces the hand-written stage cost and
// inequality rows below by the emitted tmpc_gen::cost_* / tmpc_gen::rows of a user-defined module stack.  Dynamics,
// MIRROR, the QP and everything else are shared.
#ifdef TMPC_GENERATED_STAGE
#include TMPC_GENERATED_STAGE
#endif

// The stage functions are plain per-lane arithmetic: they are compiled for the device (all kernels) and, with the same
// source, for the host -- the host instantiation exists only for tests/cpu_twin (a test-side re-run of the lane-per-trajectory
// kernel's scalar program, used to debug it against the oracle without a GPU); the product never calls it.
#define TMPC_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define TMPC_RSQ_SEED(x) __builtin_amdgcn_rsq(x)      // v_rsq_f64 / v_rcp_f64 seeds, refined by Newton steps below
#define TMPC_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#else
#define TMPC_RSQ_SEED(x) (1.0 / sqrt(x))
#define TMPC_RCP_SEED(x) (1.0 / (x))
#endif

// ---- cache policy of the read-once / write-once global streams (round-4 verdict, next-5; tools/traffic_ab.sh builds the variants) ----
// TMPC_EXP_NT = 0 (product): plain loads and stores.  1: the warm start, xinit and the output arrays -- touched once per solve -- go through
// non-temporal accesses (`nt`: streaming, low retention priority in the L2), so that they do not compete with the per-workgroup NLP workspace
// for L2 lines.  2: the parameter rows as well (read once per RTI iteration).  Measured: profiles/round5_c_traffic_nontemporal_ab.json.
#ifndef TMPC_EXP_NT
#define TMPC_EXP_NT 0
#endif
#if TMPC_EXP_NT >= 1 && defined(__HIP_DEVICE_COMPILE__)
#define TMPC_LD_IN(ptr) __builtin_nontemporal_load(ptr)
#define TMPC_ST_OUT(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define TMPC_LD_IN(ptr) (*(ptr))
#define TMPC_ST_OUT(ptr, val) (*(ptr) = (val))
#endif
#if TMPC_EXP_NT >= 2 && defined(__HIP_DEVICE_COMPILE__)
#define TMPC_LDP(ptr) __builtin_nontemporal_load(ptr)
#else
#define TMPC_LDP(ptr) (*(ptr))
#endif

namespace tmpc {

constexpr int NU = 2, NX = 5, NV = 7, NP28 = 28;
enum { ZA = 0, ZW = 1, ZX = 2, ZY = 3, ZPSI = 4, ZV = 5, ZS = 6 };

struct Dims {
    int N, S, n_lin, M, npar, n_sqp, qp_iter_max, erk_steps;
    double dt, qp_tol, reg_eps, mu0, thr0;
    double lb[NV], ub[NV];
    int n_slk;      // decomp / scenario halfspace rows  a1 x + a2 y - (b + slack) <= 0
    int slack;      // 1: slack model (nx = 6, nvar = 8 in the external layouts; MPCBase weighs the slack)
    int n_up;       // upper-bounded general rows = n_lin + n_slk.  Internal row order: [topology | slack rows | ellipsoids]
    // derived on the host (tmpc_create) so that the kernels read them as scalar kernel arguments instead of keeping
    // loop-invariant VGPR copies alive across the whole solve:
    double erk_h, erk_eta, erk_w6;   // dt / erk_steps, its half, its sixth
    double hdt2;                     // dt^2 / 2
    int model;                       // dynamics model.  0: ContouringSecondOrderUnicycleModel (solver_model.py:193-214: spline' = v), with or without the slack state;
                                     // 1: SecondOrderUnicycleModel (solver_model.py:170-191: states x, y, psi, v).  The kernels keep their 5-state layout: the fifth
                                     // slot is INERT for model 1 -- its ODE is s' = 0 (sdt = shdt2 = 0 below), no cost or row may depend on it (the generator checks),
                                     // so it decouples exactly from every other variable; callers pad xinit / x0 with 0 there.  Generated solvers only: the emitted
                                     // header fixes it (tmpc_gen::MODEL); the hand-written stage functions are the contouring stacks' (model 0).
    double sdt, shdt2;               // the spline row of [B A]: (dt, dt^2 / 2) for model 0, (0, 0) for model 1
    int dpad;                        // padding (doubles) of the stage stride of the row Jacobians in LDS (fast and compact layouts): chosen per LAUNCH by the host
                                     // from an LDS bank-conflict model of the kernel's lane map (tmpc_capi.hip pick_d_pad); layout only, no effect on results
    int prio;                        // 1: the kernel raises its wave's issue priority inside the latency-critical phases (csrc/tmpc_riccati.hpp TMPC_PRIO_*);
                                     // set per LAUNCH by the host (launch_solve): only for kernels that put two waves on every SIMD
    int cost_model;                  // 0: ContouringModule (contouring.py:48-98); 1: CurvatureAwareContouringModule (curvature_aware_contouring.py:48-105).
                                     // Kernels are instantiated per cost model (template parameter CM): this field only selects the instantiation on the host
    int row_model;                   // the M lower-bounded rows: 0: EllipsoidConstraintModule (ellipsoid_constraints.py:66-110, h >= 1); 1: GaussianConstraintModule
                                     // (gaussian_constraints.py:66-113, h >= 0; mpc_planner_jackal's default).  Compile-time in the kernels like cost_model:
                                     // the template parameter CM carries both, CM = cost_model + 2 * row_model (stage_model())
    int riccati_form;                // 0: Schur-complement recursion (every kernel); 1: square-root recursion (tmpc_riccati.hpp SQ; the SoloSqrt instantiations).
                                     // Selects the instantiation on the host only (tmpc_dims.riccati_form)
    int split_rows;                  // one-wave fast / compact kernels: the stage's rows are split over its three lanes in the linearisation and MIRROR's two blocks
                                     // run in different lanes -- decided on the HOST from the shape alone (tmpc_capi.hip split_rows_for), the same for the fast and
                                     // the compact kernel of a shape, which must stay bitwise equal (round-5 advisor: it was a layout-dependent pointer test)
};
__host__ TMPC_HD constexpr bool cm_curvature_aware(int CM) { return (CM & 1) != 0; }
__host__ TMPC_HD constexpr bool cm_gaussian_rows(int CM) { return (CM & 2) != 0; }
__host__ inline int stage_model(const Dims &d) { return d.cost_model + 2 * d.row_model; }
__host__ inline void derive_dims(Dims &d)
{
    d.n_up = d.n_lin + d.n_slk;
    d.erk_h = d.dt / d.erk_steps; d.erk_eta = 0.5 * d.erk_h; d.erk_w6 = d.erk_h / 6.0;
    d.hdt2 = 0.5 * d.dt * d.dt;
    d.sdt = d.model == 1 ? 0.0 : d.dt; d.shdt2 = d.model == 1 ? 0.0 : d.hdt2;
}
__host__ TMPC_HD int ext_nx(const Dims &d) { return NX + d.slack; }   // strides of xinit / xtraj
__host__ TMPC_HD int ext_nv(const Dims &d) { return NV + d.slack; }   // stride of x0

// packed lower-triangular index of a symmetric 7x7: (i >= j)
__host__ TMPC_HD constexpr int pidx(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ TMPC_HD constexpr int sidx(int i, int j) { return i >= j ? pidx(i, j) : pidx(j, i); }

// ---- parameter index map (reference rule: util/parameters.py:25-55, solver_definition.py:5-16) ----------
// weights: acceleration, angular_velocity, [slack,] velocity, reference_velocity, contour, lag, terminal_angle,
// terminal_contouring; then 9 per spline segment; topology rows; [ego_disc_radius, ego_disc_0_offset, 7 per ellipsoid
// (ellipsoid_constraints.py:33-49) or 6 per Gaussian obstacle (row_model 1: gaussian_constraints.py:40-52)];
// [ego_disc_0_offset unless the obstacle module defined it,] 3 per slack row (scenario rows before decomp rows).
__host__ TMPC_HD int ip_spline(const Dims &d, int seg, int which) { return 8 + d.slack + 9 * seg + which; }
__host__ TMPC_HD int ip_lin(const Dims &d, int j, int which) { return 8 + d.slack + 9 * d.S + 3 * j + which; }
__host__ TMPC_HD int ip_disc_radius(const Dims &d) { return 8 + d.slack + 9 * d.S + 3 * d.n_lin; }
__host__ TMPC_HD int ip_disc_offset(const Dims &d) { return ip_disc_radius(d) + (d.M > 0 ? 1 : 0); }
__host__ TMPC_HD int ip_ellipsoid(const Dims &d, int j, int which) { return ip_disc_radius(d) + 2 + 7 * j + which; }
__host__ TMPC_HD int ip_gauss(const Dims &d, int j, int which) { return ip_disc_radius(d) + 2 + 6 * j + which; }   // x, y, major, minor, risk, r
__host__ TMPC_HD int ip_obst_stride(const Dims &d) { return d.row_model == 1 ? 6 : 7; }
__host__ TMPC_HD int ip_slk(const Dims &d, int j, int which)
{
    return (d.M > 0 ? ip_disc_radius(d) + 2 + ip_obst_stride(d) * d.M : ip_disc_radius(d) + 1) + 3 * j + which;
}
__host__ TMPC_HD int expected_npar(const Dims &d)
{
#ifdef TMPC_GENERATED_STAGE
    (void)d;
    return tmpc_gen::NPAR;
#endif
    return 8 + d.slack + 9 * d.S + 3 * d.n_lin + (d.M > 0 ? 2 + ip_obst_stride(d) * d.M : 0) + (d.n_slk > 0 ? (d.M > 0 ? 0 : 1) + 3 * d.n_slk : 0);
}

// =============================================================================================
// Dynamics.  For xdot = [v cos psi, v sin psi, w, a, v] with inputs held constant, RK4's k2 and k3 see
// the same (psi, v), so each ERK4 sub-step is exactly a Simpson rule in the nodes
//   theta_m = psi + m*eta*w,  nu_m = v + m*eta*a,  eta = h/2,  m = 0..2*steps,
// and psi, v, spline advance exactly (polynomials of degree <= 2).  All derivatives w.r.t. (a, w, psi, v)
// follow from 12 weighted sums of cos/sin at those nodes.
// =============================================================================================
struct DynOut {
    double xn[NX];
    double Xa, Xw, Xp, Xv, Ya, Yw, Yp, Yv;                 // first derivatives of x+, y+ (others constant)
    double Xaw, Xap, Xww, Xwp, Xwv, Xpp, Xpv;              // second derivatives (Xaa = Xav = Xvv = 0)
    double Yaw, Yap, Yww, Ywp, Ywv, Ypp, Ypv;
};

TMPC_HD void dyn_eval(const Dims &d, const double *z, DynOut &o, bool second_order)
{
    const double a = z[ZA], w = z[ZW], psi = z[ZPSI], v = z[ZV];
    const double eta = d.erk_eta, w6 = d.erk_w6;
    const int last = 2 * d.erk_steps;
    double C0 = 0, C1 = 0, C2 = 0, S0 = 0, S1 = 0, S2 = 0;
    double nC0 = 0, nC1 = 0, nC2 = 0, nS0 = 0, nS1 = 0, nS2 = 0;
    // cos / sin at the equally spaced nodes theta_m = psi + m (eta w) by the angle-addition recurrence from ONE sincos(psi) and ONE sincos(eta w)
    // (round 5: seven library sincos calls per stage were ~900 of the linearisation's ~5.6 k VALU instructions; the recurrence's error grows by one
    // rounding per node -- 6 ulp at the last of the 7 nodes --, far inside what the solve's parity tolerance sees)
    double sn, cs, sd, cd;
    sincos(psi, &sn, &cs);
    sincos(eta * w, &sd, &cd);
    for (int m = 0; m <= last; m++) {
        const double om = w6 * ((m == 0 || m == last) ? 1.0 : ((m & 1) ? 4.0 : 2.0));
        const double tau = m == 0 ? 0.0 : m * eta;          // (literal 0 for the first node: nothing loop-invariant to keep in a VGPR)
        if (m > 0) { const double c1 = cs * cd - sn * sd, s1 = sn * cd + cs * sd; cs = c1; sn = s1; }
        const double nu = v + tau * a;
        const double oc = om * cs, os = om * sn;
        C0 += oc; C1 += oc * tau; C2 += oc * tau * tau;
        S0 += os; S1 += os * tau; S2 += os * tau * tau;
        nC0 += oc * nu; nC1 += oc * nu * tau; nC2 += oc * nu * tau * tau;
        nS0 += os * nu; nS1 += os * nu * tau; nS2 += os * nu * tau * tau;
    }
    o.xn[0] = z[ZX] + nC0;
    o.xn[1] = z[ZY] + nS0;
    o.xn[2] = psi + d.dt * w;
    o.xn[3] = v + d.dt * a;
    o.xn[4] = z[ZS] + d.sdt * v + d.shdt2 * a;
    o.Xa = C1; o.Xw = -nS1; o.Xp = -nS0; o.Xv = C0;
    o.Ya = S1; o.Yw = nC1; o.Yp = nC0; o.Yv = S0;
    if (second_order) {
        o.Xaw = -S2; o.Xap = -S1; o.Xww = -nC2; o.Xwp = -nC1; o.Xwv = -S1; o.Xpp = -nC0; o.Xpv = -S0;
        o.Yaw = C2; o.Yap = C1; o.Yww = -nS2; o.Ywp = -nS1; o.Ywv = C1; o.Ypp = -nS0; o.Ypv = C0;
    }
}

// [B A] (5 x 7, row-major), columns ordered like z
TMPC_HD void dyn_jacobian(const Dims &d, const DynOut &o, double *BA)
{
#pragma unroll
    for (int i = 0; i < NX * NV; i++) BA[i] = 0.0;
    BA[0 * NV + ZA] = o.Xa; BA[0 * NV + ZW] = o.Xw; BA[0 * NV + ZX] = 1.0; BA[0 * NV + ZPSI] = o.Xp; BA[0 * NV + ZV] = o.Xv;
    BA[1 * NV + ZA] = o.Ya; BA[1 * NV + ZW] = o.Yw; BA[1 * NV + ZY] = 1.0; BA[1 * NV + ZPSI] = o.Yp; BA[1 * NV + ZV] = o.Yv;
    BA[2 * NV + ZW] = d.dt; BA[2 * NV + ZPSI] = 1.0;
    BA[3 * NV + ZA] = d.dt; BA[3 * NV + ZV] = 1.0;
    BA[4 * NV + ZA] = d.shdt2; BA[4 * NV + ZV] = d.sdt; BA[4 * NV + ZS] = 1.0;
}

// W += pix * hess(x+) + piy * hess(y+)   (W full symmetric 7x7)
TMPC_HD void dyn_add_hessian(const DynOut &o, double pix, double piy, double (*W)[NV])
{
    auto add = [&](int i, int j, double val) { W[i][j] += val; W[j][i] += val; };
    add(ZA, ZW, pix * o.Xaw + piy * o.Yaw);
    add(ZA, ZPSI, pix * o.Xap + piy * o.Yap);
    W[ZW][ZW] += pix * o.Xww + piy * o.Yww;
    add(ZW, ZPSI, pix * o.Xwp + piy * o.Ywp);
    add(ZW, ZV, pix * o.Xwv + piy * o.Ywv);
    W[ZPSI][ZPSI] += pix * o.Xpp + piy * o.Ypp;
    add(ZPSI, ZV, pix * o.Xpv + piy * o.Ypv);
}

// =============================================================================================
// Cost.  One-variable second-order Taylor triples in s for everything that depends on the spline only.
// =============================================================================================
struct J1 { double v, d1, d2; };
TMPC_HD J1 j_add(J1 a, J1 b) { return {a.v + b.v, a.d1 + b.d1, a.d2 + b.d2}; }
TMPC_HD J1 j_sub(J1 a, J1 b) { return {a.v - b.v, a.d1 - b.d1, a.d2 - b.d2}; }
TMPC_HD J1 j_mul(J1 a, J1 b)
{
    return {a.v * b.v, a.v * b.d1 + a.d1 * b.v, a.v * b.d2 + 2.0 * a.d1 * b.d1 + a.d2 * b.v};
}
TMPC_HD J1 j_chain(J1 a, double f, double f1, double f2)
{
    return {f, f1 * a.d1, f1 * a.d2 + f2 * a.d1 * a.d1};
}

// spline.py:16-22 segment value / derivative as triples in s
TMPC_HD J1 seg_at(double a, double b, double c, double dd, double t)
{
    return {((a * t + b) * t + c) * t + dd, (3.0 * a * t + 2.0 * b) * t + c, 6.0 * a * t + 2.0 * b};
}
TMPC_HD J1 seg_deriv(double a, double b, double c, double t)
{
    return {(3.0 * a * t + 2.0 * b) * t + c, 6.0 * a * t + 2.0 * b, 6.0 * a};
}

#ifdef TMPC_GENERATED_STAGE
struct CostOut { double val; double g[NV]; double H[NP28]; };      // dense packed Hessian from the generated code
TMPC_HD void cost_eval(const Dims &d, const double *z, const double *p, int pstride, CostOut &o,
                                          bool derivs, double slack = 0.0)
{
    (void)d;
    if (derivs) tmpc_gen::cost_full(z, p, pstride, slack, &o.val, o.g, o.H);
    else tmpc_gen::cost_value(z, p, pstride, slack, &o.val);
}
TMPC_HD void cost_add_hessian(const CostOut &o, double scale, double (*W)[NV])
{
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const double h = scale * o.H[i * (i + 1) / 2 + j];
            W[i][j] += h;
            if (i != j) W[j][i] += h;
        }
}
#else
struct CostOut { double val; double g[NV]; double Hxx, Hxy, Hyy, Hxs, Hys, Hss, Haa, Hww, Hvv; };

// p: this stage's parameter row, element i at p[i * pstride]
// slack: the trajectory's (constant) slack value, 0 without the slack model
TMPC_HD void cost_eval(const Dims &d, const double *z, const double *p, int pstride, CostOut &o,
                                          bool derivs, double slack = 0.0)
{
    auto P = [&](int i) { return TMPC_LDP(p + (size_t)i * pstride); };
    const int ws = d.slack;
    const double w_a = P(0), w_w = P(1), w_v = P(2 + ws), v_ref = P(3 + ws), w_contour = P(4 + ws), w_lag = P(5 + ws);
    const double a = z[ZA], w = z[ZW], x = z[ZX], y = z[ZY], v = z[ZV], s = z[ZS];

    // glued spline (spline.py:28-50): value = seg[S-1]; for k = S-1..1: value = lam_k seg[k-1] + (1-lam_k) value,
    // the same blend for values and for segment derivatives.
    const int S = d.S;
    int base = ip_spline(d, S - 1, 0);
    double t = s - P(base + 8);
    J1 X = seg_at(P(base + 0), P(base + 1), P(base + 2), P(base + 3), t);
    J1 Y = seg_at(P(base + 4), P(base + 5), P(base + 6), P(base + 7), t);
    J1 DX = seg_deriv(P(base + 0), P(base + 1), P(base + 2), t);
    J1 DY = seg_deriv(P(base + 4), P(base + 5), P(base + 6), t);
    for (int k = S - 1; k >= 1; k--) {
        // lambda_k = 1/(1+exp((s - start_k + 0.02)/0.1))  (spline.py:37), overflow-safe derivatives
        const double u = (s - P(ip_spline(d, k, 8)) + 0.02) / 0.1;
        const double sig = 1.0 / (1.0 + exp(u));
        const double sig1 = -sig * (1.0 - sig);                 // d sig / du
        const double sig2 = sig1 * (2.0 * sig - 1.0);           // d2 sig / du2
        const J1 lam = {sig, sig1 * 10.0, sig2 * 100.0};
        const J1 oml = {1.0 - sig, -lam.d1, -lam.d2};
        base = ip_spline(d, k - 1, 0);
        t = s - P(base + 8);
        X = j_add(j_mul(lam, seg_at(P(base + 0), P(base + 1), P(base + 2), P(base + 3), t)), j_mul(oml, X));
        Y = j_add(j_mul(lam, seg_at(P(base + 4), P(base + 5), P(base + 6), P(base + 7), t)), j_mul(oml, Y));
        DX = j_add(j_mul(lam, seg_deriv(P(base + 0), P(base + 1), P(base + 2), t)), j_mul(oml, DX));
        DY = j_add(j_mul(lam, seg_deriv(P(base + 4), P(base + 5), P(base + 6), t)), j_mul(oml, DY));
    }
    // normalised tangent (spline.py:72-77)
    const J1 n2 = j_add(j_mul(DX, DX), j_mul(DY, DY));
    const double nrm = sqrt(n2.v);
    const J1 nr = j_chain(n2, nrm, 0.5 / nrm, -0.25 / (nrm * n2.v));
    const double rinv = 1.0 / nr.v;
    const J1 inv = j_chain(nr, rinv, -rinv * rinv, 2.0 * rinv * rinv * rinv);
    const J1 tx = j_mul(DX, inv), ty = j_mul(DY, inv);

    const double ex = x - X.v, ey = y - Y.v;
    const double ec = ty.v * ex - tx.v * ey;          // contouring.py:74
    const double el = tx.v * ex + ty.v * ey;          // contouring.py:75
    const double dv = v - v_ref;
    o.val = w_a * a * a + w_w * w * w + w_v * dv * dv + w_lag * el * el + w_contour * ec * ec;
    if (ws) o.val += P(2) * slack * slack;                  // weigh_variable("slack") (generate_jackalsimulator_solver.py:79)
    if (!derivs) return;

    // gradients / Hessians of e_c, e_l in (x, y, s)
    const double ec_x = ty.v, ec_y = -tx.v;
    const double ec_s = ty.d1 * ex - ty.v * X.d1 - tx.d1 * ey + tx.v * Y.d1;
    const double ec_xs = ty.d1, ec_ys = -tx.d1;
    const double ec_ss = ty.d2 * ex - 2.0 * ty.d1 * X.d1 - ty.v * X.d2 - tx.d2 * ey + 2.0 * tx.d1 * Y.d1 + tx.v * Y.d2;
    const double el_x = tx.v, el_y = ty.v;
    const double el_s = tx.d1 * ex - tx.v * X.d1 + ty.d1 * ey - ty.v * Y.d1;
    const double el_xs = tx.d1, el_ys = ty.d1;
    const double el_ss = tx.d2 * ex - 2.0 * tx.d1 * X.d1 - tx.v * X.d2 + ty.d2 * ey - 2.0 * ty.d1 * Y.d1 - ty.v * Y.d2;

    const double cl = 2.0 * w_lag, cc = 2.0 * w_contour;
    o.g[ZA] = 2.0 * w_a * a; o.g[ZW] = 2.0 * w_w * w; o.g[ZPSI] = 0.0; o.g[ZV] = 2.0 * w_v * dv;
    o.g[ZX] = cl * el * el_x + cc * ec * ec_x;
    o.g[ZY] = cl * el * el_y + cc * ec * ec_y;
    o.g[ZS] = cl * el * el_s + cc * ec * ec_s;
    o.Haa = 2.0 * w_a; o.Hww = 2.0 * w_w; o.Hvv = 2.0 * w_v;
    o.Hxx = cl * el_x * el_x + cc * ec_x * ec_x;
    o.Hxy = cl * el_x * el_y + cc * ec_x * ec_y;
    o.Hyy = cl * el_y * el_y + cc * ec_y * ec_y;
    o.Hxs = cl * (el_x * el_s + el * el_xs) + cc * (ec_x * ec_s + ec * ec_xs);
    o.Hys = cl * (el_y * el_s + el * el_ys) + cc * (ec_y * ec_s + ec * ec_ys);
    o.Hss = cl * (el_s * el_s + el * el_ss) + cc * (ec_s * ec_s + ec * ec_ss);
}

// W += scale * hess(cost)
TMPC_HD void cost_add_hessian(const CostOut &o, double scale, double (*W)[NV])
{
    W[ZA][ZA] += scale * o.Haa; W[ZW][ZW] += scale * o.Hww; W[ZV][ZV] += scale * o.Hvv;
    W[ZX][ZX] += scale * o.Hxx; W[ZY][ZY] += scale * o.Hyy; W[ZS][ZS] += scale * o.Hss;
    W[ZX][ZY] += scale * o.Hxy; W[ZY][ZX] += scale * o.Hxy;
    W[ZX][ZS] += scale * o.Hxs; W[ZS][ZX] += scale * o.Hxs;
    W[ZY][ZS] += scale * o.Hys; W[ZS][ZY] += scale * o.Hys;
}

// ---- curvature-aware contouring (BASELINE configs[2] "CA-MPC") ------------------------------------------------------------
// CurvatureAwareContouringObjective.get_value (curvature_aware_contouring.py:48-105) at stage_idx = 1, on top of the MPCBase terms:
//   cost = w_a a^2 + w_w w^2 + w_v (v - v_ref)^2 [+ w_slack slack^2]                      mpc_base.py:47-60
//        + w_contour ((x - X)^2 + (y - Y)^2)                                               :87-89  (no projection on the tangent frame)
//        + w_v (s_dot - v_ref)^2,   s_dot = v (cos psi tx + sin psi ty) / q,               :82-84,90
//          q = 1 - ((x - X) X'' + (y - Y) Y''),   X'' / Y'' = Spline.deriv2 (spline.py:52-56: the lambda blend of the segments' second derivatives)
// Same parameter map as the MPCC stack (the module's C++ side leaves velocity / reference_velocity to MPCBaseModule,
// curvature_aware_contouring.cpp:15-49; `lag` is defined but unused, :26).  The model keeps s' = v (route 1: the reference's
// own CA model class needs Forces-style discrete dynamics and is rejected by its acados path, solver_model.py:217-221).
// s_dot couples (psi, v) with (x, y, s): the cost Hessian is dense on those five variables, and MIRROR sees a full 7 x 7 matrix (mirror7's
// coupled branch).  H: packed lower over (x, y, psi, v, s) = z[2..6].
TMPC_HD J1 seg_deriv2(double a, double b, double t) { return {6.0 * a * t + 2.0 * b, 6.0 * a, 0.0}; }

struct CostOutCA { double val; double g[NV]; double Haa, Hww; double H[15]; };

TMPC_HD void cost_eval_ca(const Dims &d, const double *z, const double *p, int pstride, CostOutCA &o, bool derivs, double slack = 0.0)
{
    auto P = [&](int i) { return TMPC_LDP(p + (size_t)i * pstride); };
    const int ws = d.slack;
    const double w_a = P(0), w_w = P(1), w_v = P(2 + ws), v_ref = P(3 + ws), w_contour = P(4 + ws);
    const double a = z[ZA], w = z[ZW], x = z[ZX], y = z[ZY], psi = z[ZPSI], v = z[ZV], s = z[ZS];
    const int S = d.S;
    int base = ip_spline(d, S - 1, 0);
    double t = s - P(base + 8);
    J1 X = seg_at(P(base + 0), P(base + 1), P(base + 2), P(base + 3), t);
    J1 Y = seg_at(P(base + 4), P(base + 5), P(base + 6), P(base + 7), t);
    J1 DX = seg_deriv(P(base + 0), P(base + 1), P(base + 2), t);
    J1 DY = seg_deriv(P(base + 4), P(base + 5), P(base + 6), t);
    J1 XPP = seg_deriv2(P(base + 0), P(base + 1), t);
    J1 YPP = seg_deriv2(P(base + 4), P(base + 5), t);
    for (int k = S - 1; k >= 1; k--) {
        const double u = (s - P(ip_spline(d, k, 8)) + 0.02) / 0.1;
        const double sig = 1.0 / (1.0 + exp(u));
        const double sig1 = -sig * (1.0 - sig);
        const double sig2 = sig1 * (2.0 * sig - 1.0);
        const J1 lam = {sig, sig1 * 10.0, sig2 * 100.0};
        const J1 oml = {1.0 - sig, -lam.d1, -lam.d2};
        base = ip_spline(d, k - 1, 0);
        t = s - P(base + 8);
        X = j_add(j_mul(lam, seg_at(P(base + 0), P(base + 1), P(base + 2), P(base + 3), t)), j_mul(oml, X));
        Y = j_add(j_mul(lam, seg_at(P(base + 4), P(base + 5), P(base + 6), P(base + 7), t)), j_mul(oml, Y));
        DX = j_add(j_mul(lam, seg_deriv(P(base + 0), P(base + 1), P(base + 2), t)), j_mul(oml, DX));
        DY = j_add(j_mul(lam, seg_deriv(P(base + 4), P(base + 5), P(base + 6), t)), j_mul(oml, DY));
        XPP = j_add(j_mul(lam, seg_deriv2(P(base + 0), P(base + 1), t)), j_mul(oml, XPP));
        YPP = j_add(j_mul(lam, seg_deriv2(P(base + 4), P(base + 5), t)), j_mul(oml, YPP));
    }
    const J1 n2 = j_add(j_mul(DX, DX), j_mul(DY, DY));
    const double nrm = sqrt(n2.v);
    const J1 nr = j_chain(n2, nrm, 0.5 / nrm, -0.25 / (nrm * n2.v));
    const double rinv = 1.0 / nr.v;
    const J1 inv = j_chain(nr, rinv, -rinv * rinv, 2.0 * rinv * rinv * rinv);
    const J1 tx = j_mul(DX, inv), ty = j_mul(DY, inv);

    const double ex = x - X.v, ey = y - Y.v;
    const double q = 1.0 - (ex * XPP.v + ey * YPP.v);
    const double rho = 1.0 / q;
    double sp, cp;
    sincos(psi, &sp, &cp);
    const double c = cp * tx.v + sp * ty.v;
    const double f = v * c * rho;                           // s_dot
    const double df = f - v_ref, dv = v - v_ref;
    o.val = w_a * a * a + w_w * w * w + w_v * dv * dv + w_contour * (ex * ex + ey * ey) + w_v * df * df;
    if (ws) o.val += P(2) * slack * slack;
    if (!derivs) return;

    // q and rho = 1 / q in (x, y, s)
    const double q_x = -XPP.v, q_y = -YPP.v;
    const double q_s = X.d1 * XPP.v - ex * XPP.d1 + Y.d1 * YPP.v - ey * YPP.d1;
    const double q_xs = -XPP.d1, q_ys = -YPP.d1;
    const double q_ss = X.d2 * XPP.v + 2.0 * X.d1 * XPP.d1 - ex * XPP.d2 + Y.d2 * YPP.v + 2.0 * Y.d1 * YPP.d1 - ey * YPP.d2;
    const double rho2 = rho * rho, rho3 = rho2 * rho;
    const double r_x = -q_x * rho2, r_y = -q_y * rho2, r_s = -q_s * rho2;
    const double r_xx = 2.0 * q_x * q_x * rho3, r_xy = 2.0 * q_x * q_y * rho3, r_yy = 2.0 * q_y * q_y * rho3;
    const double r_xs = -q_xs * rho2 + 2.0 * q_x * q_s * rho3, r_ys = -q_ys * rho2 + 2.0 * q_y * q_s * rho3;
    const double r_ss = -q_ss * rho2 + 2.0 * q_s * q_s * rho3;
    // c = cos psi tx + sin psi ty in (psi, s)
    const double c_p = -sp * tx.v + cp * ty.v, c_s = cp * tx.d1 + sp * ty.d1;
    const double c_ps = -sp * tx.d1 + cp * ty.d1, c_ss = cp * tx.d2 + sp * ty.d2;      // (c_pp = -c)
    // f = v c rho: gradient and packed lower Hessian over (x, y, psi, v, s)
    double fg[5], fH[15];
    fg[0] = v * c * r_x; fg[1] = v * c * r_y; fg[2] = v * c_p * rho; fg[3] = c * rho; fg[4] = v * (c_s * rho + c * r_s);
    fH[0] = v * c * r_xx;                                                               // xx
    fH[1] = v * c * r_xy; fH[2] = v * c * r_yy;                                         // yx yy
    fH[3] = v * c_p * r_x; fH[4] = v * c_p * r_y; fH[5] = -v * c * rho;                 // px py pp
    fH[6] = c * r_x; fH[7] = c * r_y; fH[8] = c_p * rho; fH[9] = 0.0;                   // vx vy vp vv
    fH[10] = v * (c_s * r_x + c * r_xs); fH[11] = v * (c_s * r_y + c * r_ys);           // sx sy
    fH[12] = v * (c_ps * rho + c_p * r_s); fH[13] = c_s * rho + c * r_s;                // sp sv
    fH[14] = v * (c_ss * rho + 2.0 * c_s * r_s + c * r_ss);                             // ss
    const double cv2 = 2.0 * w_v, cc = 2.0 * w_contour;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) o.H[i * (i + 1) / 2 + j] = cv2 * (fg[i] * fg[j] + df * fH[i * (i + 1) / 2 + j]);
    o.g[ZA] = 2.0 * w_a * a; o.g[ZW] = 2.0 * w_w * w;
    o.g[ZX] = cv2 * df * fg[0] + cc * ex;
    o.g[ZY] = cv2 * df * fg[1] + cc * ey;
    o.g[ZPSI] = cv2 * df * fg[2];
    o.g[ZV] = cv2 * df * fg[3] + cv2 * dv;
    o.g[ZS] = cv2 * df * fg[4] - cc * (ex * X.d1 + ey * Y.d1);
    o.Haa = 2.0 * w_a; o.Hww = 2.0 * w_w;
    o.H[0] += cc; o.H[2] += cc; o.H[9] += cv2;                                          // xx, yy (distance), vv (MPCBase velocity term)
    o.H[10] -= cc * X.d1; o.H[11] -= cc * Y.d1;                                         // sx, sy
    o.H[14] += cc * (X.d1 * X.d1 + Y.d1 * Y.d1 - ex * X.d2 - ey * Y.d2);                // ss
}

TMPC_HD void cost_add_hessian_ca(const CostOutCA &o, double scale, double (*W)[NV])
{
    W[ZA][ZA] += scale * o.Haa; W[ZW][ZW] += scale * o.Hww;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const double h = scale * o.H[i * (i + 1) / 2 + j];
            W[ZX + i][ZX + j] += h;
            if (i != j) W[ZX + j][ZX + i] += h;
        }
}
#endif  // TMPC_GENERATED_STAGE (hand-written cost)

// =============================================================================================
// Inequality rows.  Row r < n_lin: topology halfspace a1 x + a2 y - b  (<= 0);
// row n_lin + j: ellipsoid  dlt^T R(po)^T diag(1/(maj sqrt(chi)+rd+r)^2, 1/(min sqrt(chi)+rd+r)^2) R(po) dlt (>= 1),
// dlt = [x + off cos psi - ox, y + off sin psi - oy].  Only (x, y, psi) carry derivatives.
// =============================================================================================
struct RowOut { double h; double gx, gy, gp; double Hxx, Hxy, Hyy, Hxp, Hyp, Hpp; };

TMPC_HD void lin_row_eval(const Dims &d, const double *z, const double *p, int pstride, int j, RowOut &o)
{
    const double a1 = TMPC_LDP(p + (size_t)ip_lin(d, j, 0) * pstride), a2 = TMPC_LDP(p + (size_t)ip_lin(d, j, 1) * pstride);
    const double b = TMPC_LDP(p + (size_t)ip_lin(d, j, 2) * pstride);
    o.h = a1 * z[ZX] + a2 * z[ZY] - b;
    o.gx = a1; o.gy = a2; o.gp = 0.0;
    o.Hxx = o.Hxy = o.Hyy = o.Hxp = o.Hyp = o.Hpp = 0.0;
}

TMPC_HD void ellipsoid_row_eval(const Dims &d, const double *z, const double *p, int pstride, int j,
                                                   double r_disc, double off, double spsi, double cpsi, RowOut &o)
{
    auto P = [&](int w) { return TMPC_LDP(p + (size_t)ip_ellipsoid(d, j, w) * pstride); };
    const double ox = P(0), oy = P(1), opsi = P(2), chi = P(5), r = P(6);
    const double sq = sqrt(chi);
    const double major = P(3) * sq, minor = P(4) * sq;                    // ellipsoid_constraints.py:94-95
    const double ra = major + r_disc + r, rb = minor + r_disc + r;
    const double ab00 = 1.0 / (ra * ra), ab11 = 1.0 / (rb * rb);           // :97,100
    double so, co;
    sincos(opsi, &so, &co);
    const double m00 = co * co * ab00 + so * so * ab11;                    // R^T ab R, R = rotation_matrix(opsi)
    const double m01 = co * so * (ab11 - ab00);
    const double m11 = so * so * ab00 + co * co * ab11;
    const double px = z[ZX] + off * cpsi - ox, py = z[ZY] + off * spsi - oy;
    const double qx = -off * spsi, qy = off * cpsi;                        // d(px,py)/dpsi
    const double gx = 2.0 * (m00 * px + m01 * py), gy = 2.0 * (m01 * px + m11 * py);
    o.h = m00 * px * px + 2.0 * m01 * px * py + m11 * py * py;
    o.gx = gx; o.gy = gy; o.gp = gx * qx + gy * qy;
    o.Hxx = 2.0 * m00; o.Hxy = 2.0 * m01; o.Hyy = 2.0 * m11;
    o.Hxp = 2.0 * (m00 * qx + m01 * qy);
    o.Hyp = 2.0 * (m01 * qx + m11 * qy);
    o.Hpp = 2.0 * (m00 * qx * qx + 2.0 * m01 * qx * qy + m11 * qy * qy) - gx * qy + gy * qx;
}

// Gaussian chance constraint (gaussian_constraints.py:66-113; lower bound 0, :54-58):
//   h = a . d - (r_disc + r) - erfinv(1 - 2 risk) sqrt(2 a^T diag(major^2, minor^2) a),   d = disc position - obstacle mean,  a = d / |d|
// erfinv by the script's rational start and two Newton steps (:103-111); it depends on the parameters only.
TMPC_HD double gauss_quantile(double risk)
{
    const double xe = 1.0 - 2.0 * risk;
    const double zz = sqrt(-log((1.0 - xe) / 2.0));
    double ye = (((1.641345311 * zz + 3.429567803) * zz - 1.624906493) * zz - 1.970840454) / ((1.637067800 * zz + 3.543889200) * zz + 1.0);
    for (int it = 0; it < 2; it++) ye = ye - (erf(ye) - xe) / (1.1283791670955126 * exp(-ye * ye));       // 2 / sqrt(pi)
    return ye;
}
// With q = |d|^2, u = (major^2 dx^2 + minor^2 dy^2) / q (the Rayleigh quotient of a) and f = sqrt(2 u):  h = sqrt(q) - R - ye f, and
//   u_x = 2 dx (major^2 - u) / q,  u_xx = 2 (major^2 - u) / q - 8 dx^2 (major^2 - u) / q^2,  u_xy = -4 dx dy (major^2 + minor^2 - 2 u) / q^2,
//   f f_x = u_x,  f f_xx = u_xx - f_x^2,  f f_xy = u_xy - f_x f_y;  psi enters through the disc offset exactly as in the ellipsoid row.
TMPC_HD void gauss_row_eval(const Dims &d, const double *z, const double *p, int pstride, int j,
                            double r_disc, double off, double spsi, double cpsi, double ye, RowOut &o)
{
    auto P = [&](int w) { return TMPC_LDP(p + (size_t)ip_gauss(d, j, w) * pstride); };
    const double ox = P(0), oy = P(1), s0 = P(2) * P(2), s1 = P(3) * P(3), R = r_disc + P(5);
    const double px = z[ZX] + off * cpsi - ox, py = z[ZY] + off * spsi - oy;
    const double qx = -off * spsi, qy = off * cpsi;                        // d(px,py)/dpsi
    const double q = px * px + py * py, iq = 1.0 / q, rho = sqrt(q), ir = rho * iq, ir3 = ir * iq;
    const double u = (s0 * px * px + s1 * py * py) * iq, A = s0 - u, B = s1 - u;
    const double f = sqrt(2.0 * u), jf = 1.0 / f;
    const double ux = 2.0 * px * A * iq, uy = 2.0 * py * B * iq;
    const double uxx = 2.0 * A * iq - 8.0 * px * px * A * iq * iq, uyy = 2.0 * B * iq - 8.0 * py * py * B * iq * iq;
    const double uxy = -4.0 * px * py * (A + B) * iq * iq;
    const double fx = ux * jf, fy = uy * jf;
    const double fxx = (uxx - fx * fx) * jf, fxy = (uxy - fx * fy) * jf, fyy = (uyy - fy * fy) * jf;
    const double gx = px * ir - ye * fx, gy = py * ir - ye * fy;
    const double hxx = py * py * ir3 - ye * fxx, hxy = -px * py * ir3 - ye * fxy, hyy = px * px * ir3 - ye * fyy;
    o.h = rho - R - ye * f;
    o.gx = gx; o.gy = gy; o.gp = gx * qx + gy * qy;
    o.Hxx = hxx; o.Hxy = hxy; o.Hyy = hyy;
    o.Hxp = hxx * qx + hxy * qy;
    o.Hyp = hxy * qx + hyy * qy;
    o.Hpp = (hxx * qx * qx + 2.0 * hxy * qx * qy + hyy * qy * qy) - gx * qy + gy * qx;
}

// decomp / scenario halfspace (decomp_constraints.py:86-96, scenario_constraints.py:82-92):
//   a1 (x + off cos psi) + a2 (y + off sin psi) - (b + slack)  <= 0
TMPC_HD void slk_row_eval(const Dims &d, const double *z, const double *p, int pstride, int j, double off,
                                             double spsi, double cpsi, double slack, RowOut &o)
{
    const double a1 = TMPC_LDP(p + (size_t)ip_slk(d, j, 0) * pstride), a2 = TMPC_LDP(p + (size_t)ip_slk(d, j, 1) * pstride);
    const double b = TMPC_LDP(p + (size_t)ip_slk(d, j, 2) * pstride);
    o.h = a1 * (z[ZX] + off * cpsi) + a2 * (z[ZY] + off * spsi) - (b + slack);
    o.gx = a1; o.gy = a2; o.gp = off * (a2 * cpsi - a1 * spsi);
    o.Hxx = o.Hxy = o.Hyy = o.Hxp = o.Hyp = 0.0;
    o.Hpp = -off * (a1 * cpsi + a2 * spsi);
}

TMPC_HD void row_add_hessian(const RowOut &o, double scale, double (*W)[NV])
{
    W[ZX][ZX] += scale * o.Hxx; W[ZY][ZY] += scale * o.Hyy; W[ZPSI][ZPSI] += scale * o.Hpp;
    W[ZX][ZY] += scale * o.Hxy; W[ZY][ZX] += scale * o.Hxy;
    W[ZX][ZPSI] += scale * o.Hxp; W[ZPSI][ZX] += scale * o.Hxp;
    W[ZY][ZPSI] += scale * o.Hyp; W[ZPSI][ZY] += scale * o.Hyp;
}

// =============================================================================================
// MIRROR: W <- V max(|e|, eps) V^T via cyclic Jacobi, all in registers (fully unrolled 7x7).
// Only the reconstructed matrix leaves this function.
// =============================================================================================
// 1/sqrt(x), 1/x for x > 0: hardware seed (~5e-8) + one third-order step (error of order e^3: full double precision; five / three
// dependent operations instead of the eight / four of two Newton steps, and far fewer than the IEEE sqrt / divide sequences)
TMPC_HD double st_rsqrt(double x)
{
    const double y = TMPC_RSQ_SEED(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}
TMPC_HD double st_rcp(double x)
{
    const double y = TMPC_RCP_SEED(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

// MIRROR of an n x n symmetric matrix held in registers (fully unrolled cyclic Jacobi): A <- V max(|e|, eps) V^T.
// A sweep starts while the squared off-diagonal norm exceeds TMPC_MIRROR_TOL2 times the squared Frobenius norm (1e-32: the oracle's; looser
// thresholds were measured in round 5 and buy 0.1-0.2 %: profiles/round5_h_mirror_tol_ab.jsonl -- not taken).
#ifndef TMPC_MIRROR_TOL2
#define TMPC_MIRROR_TOL2 1e-32
#endif
// One Jacobi rotation's parameters from the pivot block (a_pp, a_qq, a_pq): c = cos, s = sin, t = tan of the angle |theta| <= pi / 4 that annihilates a_pq:
// t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (a_qq - a_pp) / (2 a_pq), written division-free:
// t = a_pq sgn(tau) / (|tau| + sqrt(tau^2 + a_pq^2)), tau = (a_qq - a_pp) / 2;  c = 1 / sqrt(t^2 + 1), s = t c.
// (Round 6 measured a form with TWO reciprocal square roots on the dependent chain instead of rsqrt -> reciprocal -> rsqrt -- cos^2 = (1 + |tau| / h) / 2 =: x,
// c = x rsqrt(x), s = g rsqrt(x), t = g rsqrt(x)^2, g = sgn(tau) a_pq / (2 h): -DTMPC_EXP_ROT2 -- the same rotation to rounding, every parity test green, and
// 0.4 % SLOWER on the saturated compact kernel, +-0.5 % on the ticks: profiles/round6_mirror_rotation_ab.jsonl.  Not taken.)
// `live` false (|a_pq| <= 1e-150): the identity (c = 1, s = t = 0) -- by selects, for the branch-free callers.
struct JacobiRot { double c, s, t; };
template <bool SELECT>
TMPC_HD JacobiRot jacobi_rot(double app, double aqq, double apq, bool live)
{
    const double tau = 0.5 * (aqq - app);
    double h2 = tau * tau + apq * apq;
    if (SELECT) h2 = live ? h2 : 1.0;                       // (keeps the reciprocal square root finite)
    JacobiRot r;
#ifndef TMPC_EXP_ROT2
    const double hyp = h2 * st_rsqrt(h2);
    r.t = (tau >= 0.0 ? apq : -apq) * st_rcp(fabs(tau) + hyp);
    r.c = st_rsqrt(r.t * r.t + 1.0); r.s = r.t * r.c;
#else
    const double rh = st_rsqrt(h2);
    const double g = (tau >= 0.0 ? 0.5 : -0.5) * apq * rh;
    const double x = fma(0.5 * fabs(tau), rh, 0.5);
    const double rc = st_rsqrt(x);
    r.c = x * rc; r.s = g * rc; r.t = r.s * rc;
#endif
    if (SELECT) { r.c = live ? r.c : 1.0; r.s = live ? r.s : 0.0; r.t = live ? r.t : 0.0; }
    return r;
}
// A <- G^T A G on the symmetric matrix, V <- V G: the pivot block in closed form (a_pp - t a_pq, a_qq + t a_pq, 0), the other rows' (p, q) entries once and
// mirrored -- (NN - 2) pairs where the two full passes over columns and rows (round 1-3) took 2 NN: a third of the rotation's arithmetic (round 4)
template <int NN>
TMPC_HD void jacobi_apply(double (&A)[NN][NN], double (&V)[NN][NN], int p, int q, const JacobiRot &r, double apq, double apq_after)
{
    const double c = r.c, s = r.s;
    A[p][p] -= r.t * apq; A[q][q] += r.t * apq; A[p][q] = apq_after; A[q][p] = apq_after;
#pragma unroll
    for (int k = 0; k < NN; k++) {
        if (k == p || k == q) continue;
        const double akp = A[k][p], akq = A[k][q];
        const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
        A[k][p] = np_; A[p][k] = np_; A[k][q] = nq_; A[q][k] = nq_;
    }
#pragma unroll
    for (int k = 0; k < NN; k++) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
    }
}
// PAIR (4 x 4 only; the latency kernels, where one wave per SIMD leaves a rotation's dependent chain exposed): the sweep visits the pivots in the round-robin
// order (0,1)(2,3) | (0,2)(1,3) | (0,3)(1,2) -- the two rotations of a round touch disjoint index pairs, so the second one's parameters do not depend on the
// first one's update and both chains are computed side by side, branch-free; applied one after the other.  Another (equally valid) cyclic ordering than
// the row-wise one: the same limit V |e| V^T, to rounding.
template <int NN, bool PAIR = false>
TMPC_HD void mirror_n(double (&A)[NN][NN], double eps)
{
    static_assert(!PAIR || NN == 4, "paired sweep: 4 x 4");
    double V[NN][NN];
#pragma unroll
    for (int i = 0; i < NN; i++)
#pragma unroll
        for (int j = 0; j < NN; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < NN; i++) {
            dg += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < NN; j++) off += A[i][j] * A[i][j];
        }
        if (off <= TMPC_MIRROR_TOL2 * (dg + off) || off == 0.0) break;
        if constexpr (PAIR) {
            constexpr int RR[3][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}};
#pragma unroll
            for (int rd = 0; rd < 3; rd++) {
                const int p = RR[rd][0], q = RR[rd][1], u = RR[rd][2], w = RR[rd][3];
                const double apq = A[p][q], auw = A[u][w];
                const bool l1 = fabs(apq) > 1e-150, l2 = fabs(auw) > 1e-150;
                const JacobiRot r1 = jacobi_rot<true>(A[p][p], A[q][q], apq, l1), r2 = jacobi_rot<true>(A[u][u], A[w][w], auw, l2);
                jacobi_apply<NN>(A, V, p, q, r1, apq, l1 ? 0.0 : apq);
                jacobi_apply<NN>(A, V, u, w, r2, auw, l2 ? 0.0 : auw);
            }
        } else {
#pragma unroll
        for (int p = 0; p < NN - 1; p++) {
#pragma unroll
            for (int q = p + 1; q < NN; q++) {
                const double apq = A[p][q];
                // (a branch, not an identity rotation: the joins cost ~25 register copies per rotation, a third of MIRROR's instructions, and a
                // branch-free form -- t = 0, c = 1 by selects -- removes them; measured in round 5 it is 3.6 % faster on a lone wave and 0.4 % SLOWER
                // on the saturated compact kernel, profiles/round5_m_factor_unroll_rotation_ab.jsonl: not taken)
                if (fabs(apq) > 1e-150) {                       // (also keeps tau^2 + apq^2 away from underflow)
                    const JacobiRot r = jacobi_rot<false>(A[p][p], A[q][q], apq, true);
                    jacobi_apply<NN>(A, V, p, q, r, apq, 0.0);
                }
            }
        }
        }
    }
    double e[NN];
#pragma unroll
    for (int i = 0; i < NN; i++) {
        double ei = A[i][i];
        if (ei >= -eps && ei <= eps) ei = eps; else if (ei < 0.0) ei = -ei;
        e[i] = ei;
    }
#pragma unroll
    for (int i = 0; i < NN; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NN; k++) acc += V[i][k] * e[k] * V[j][k];
            A[i][j] = acc; A[j][i] = acc;
        }
}

// MIRROR of the 7x7 stage Hessian.  With a zero disc offset (Jackal: one disc at the centre, data_preparation.cpp:25-28)
// the Lagrangian Hessian is block diagonal under the permutation {a, w, psi, v} | {x, y, spline} (dynamics curvature
// couples the first set, contouring cost + ellipsoids the second): the two blocks are then regularised separately
// (same result, ~half the rotations' work); any coupling entry != 0 falls back to the full 7x7 iteration.
TMPC_HD void mirror7(double (*A)[NV], double eps)
{
    constexpr int IA[4] = {ZA, ZW, ZPSI, ZV}, IB[3] = {ZX, ZY, ZS};
    bool coupled = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) coupled |= (A[IA[i]][IB[j]] != 0.0) | (A[IB[j]][IA[i]] != 0.0);
    if (!coupled) {
        double Ba[4][4], Bb[3][3];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) Ba[i][j] = A[IA[i]][IA[j]];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Bb[i][j] = A[IB[i]][IB[j]];
        mirror_n<4>(Ba, eps);
        mirror_n<3>(Bb, eps);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) A[IA[i]][IA[j]] = Ba[i][j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) A[IB[i]][IB[j]] = Bb[i][j];
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

// Lagrangian Hessian of one stage (before MIRROR): dt*hess(l) + pi_x hess(x+) + pi_y hess(y+) + sum_r lamh_r hess(h_r),
// plus the linearisation data of the stage.  lamh(r) is supplied by a functor (zero for inactive rows).
// Rows are numbered in the kernels' internal order [topology | slack rows | ellipsoids] (upper-bounded rows first).
struct EllAll { TMPC_HD int operator()() const { return 0; } };      // default obstacle-row share: start at the first row (with ell_step = 1: all of them)
template <int CM = 0, typename LamH, typename RowSink, typename EllFirst = EllAll>
TMPC_HD void stage_linearise(const Dims &d, const double *z, const double *p, int pstride,
                                                double pix, double piy, LamH lamh, RowSink sink,
                                                double (*W)[NV], double *g, double *BA, double *xn, double slack = 0.0,
                                                double *stash = nullptr, long long own_delta = 0, int part = 0,
                                                EllFirst ell_first = EllFirst(), int ell_step = 1, bool rows_only = false)
{
    // ell_first / ell_step / rows_only (round 5, the one-wave kernels): this lane evaluates every ell_step-th row of each row class (halfspace, scenario /
    // decomp, obstacle rows), starting at ell_first() (a functor, called where a loop starts: its value is not held in a register across the stage);
    // with rows_only everything ELSE that enters W (dynamics curvature, cost Hessian) is multiplied by zero, so that
    // W comes back as exactly this lane's share of the rows' Hessian -- three lanes of a stage take a third of the rows each (linearise,
    // tmpc_kernels.hpp) and the owner adds the helpers' shares to its W.  (A lane mask, turned into 0.0 / 1.0 where it is used: no register held
    // across the stage.)  Defaults: every row, everything in W.
    // part (hand-written stages; compile-time at every call site): 0 = the whole stage; 1 = dynamics and the first half of the ellipsoid rows
    // (W = their share of the Lagrangian Hessian, BA, xn; g untouched); 2 = cost, topology and scenario / decomp rows, the other ellipsoid
    // rows (W = their share, g; BA, xn untouched).  The two-wave kernels run 1 and 2 on different waves at the same time and add the two W
    // (linearise, tmpc_solve.hip).  Measured shares of a stage's 46 k cycles (cfg 2): dynamics 5.5 k, eight ellipsoid rows 10.2 k, cost and
    // halfspace rows 8.3 k -- and 22.3 k for MIRROR, which needs the complete W and stays on one wave.
    // Round 6, the four-wave kernels: 3 = dynamics alone; 4 = cost, topology and scenario / decomp rows (no obstacle rows); 5 = obstacle rows alone
    // (this lane's share of them: ell_first / ell_step; call with rows_only); 6 = the cost alone; 7 = every row class alone (this lane's share of each:
    // ell_first / ell_step; call with rows_only) -- the four-wave linearisation runs 3 | 6 | 7 | 7: the halfspace and scenario rows leave the cost's wave.
    // own_delta (doubles, wave-uniform): distance from `p` to the trajectory's OWN parameter row when `p` is a row it shares with others
    // (tmpc_set_param_sharing): the topology and scenario halfspaces (ip_lin / ip_slk) are read from p + own_delta, everything else
    // from p.  A delta, not a second pointer: the second address then lives only across the halfspace loads (the linearisation is the
    // register-hungriest phase of the compact kernels: a second per-lane pointer cost them 8 B of scratch).
#ifdef TMPC_GENERATED_STAGE
    // The emitted cost is long straight-line code: evaluate it first and park dt * Hessian in `stash` (28 doubles of LDS, the
    // stage's own W slot), so that its temporaries are dead before the dynamics / rows / W accumulation start.
    CostOut co;
    cost_eval(d, z, p, pstride, co, true, slack);
#pragma unroll
    for (int i = 0; i < NV; i++) g[i] = d.dt * co.g[i];          // stage cost scaled by the shooting interval
    if (stash) {
#pragma unroll
        for (int e = 0; e < NP28; e++) stash[e] = d.dt * co.H[e];
    }
#endif
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int j = 0; j < NV; j++) W[i][j] = 0.0;
#ifndef TMPC_GENERATED_STAGE
    if (part != 2 && part != 4 && part != 5 && part != 6 && part != 7)
#endif
    {
        DynOut dy;
        dyn_eval(d, z, dy, true);
        dyn_jacobian(d, dy, BA);
#pragma unroll
        for (int i = 0; i < NX; i++) xn[i] = dy.xn[i];
        dyn_add_hessian(dy, rows_only ? 0.0 : pix, rows_only ? 0.0 : piy, W);
    }
#ifdef TMPC_GENERATED_STAGE
    if (stash) {
#pragma unroll
        for (int i = 0; i < NV; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                const double h = stash[i * (i + 1) / 2 + j];
                W[i][j] += h;
                if (i != j) W[j][i] += h;
            }
    } else {
        cost_add_hessian(co, d.dt, W);
    }
    // generated rows, all normalised to g(z) <= 0 (internal order = emitted order, every row upper-bounded)
    tmpc_gen::rows(z, p, pstride, slack, [&](int k, double h, double gx, double gy, double gp, double hxx, double hxy, double hyy,
                                             double hxp, double hyp, double hpp) {
        RowOut ro;
        ro.h = h; ro.gx = gx; ro.gy = gy; ro.gp = gp;
        ro.Hxx = hxx; ro.Hxy = hxy; ro.Hyy = hyy; ro.Hxp = hxp; ro.Hyp = hyp; ro.Hpp = hpp;
        row_add_hessian(ro, lamh(k), W);
        sink(k, ro);
    });
#else
    RowOut ro;
    if (part == 3) return;
    if (part != 1 && part != 5 && part != 7) {
        if constexpr (cm_curvature_aware(CM)) {
            CostOutCA co;
            cost_eval_ca(d, z, p, pstride, co, true, slack);
#pragma unroll
            for (int i = 0; i < NV; i++) g[i] = d.dt * co.g[i];
            cost_add_hessian_ca(co, rows_only ? 0.0 : d.dt, W);
        } else {
        CostOut co;
        cost_eval(d, z, p, pstride, co, true, slack);
#pragma unroll
        for (int i = 0; i < NV; i++) g[i] = (i == ZPSI) ? 0.0 : d.dt * co.g[i];   // stage cost scaled by the shooting interval (the cost
                                                                            // does not depend on psi: literal 0, not a hoisted dt * 0)
        cost_add_hessian(co, rows_only ? 0.0 : d.dt, W);
        }
    }
    if (part == 6) return;
    if (part != 1 && part != 5) {
        for (int j = ell_first(); j < d.n_lin; j += ell_step) {   // (halfspace rows have no curvature: nothing of them enters W)
            lin_row_eval(d, z, p + own_delta, pstride, j, ro);
            sink(j, ro);
        }
    }
    if (d.M == 0 && d.n_slk == 0) return;
    const double r_disc = d.M > 0 ? TMPC_LDP(p + (size_t)ip_disc_radius(d) * pstride) : 0.0, off = TMPC_LDP(p + (size_t)ip_disc_offset(d) * pstride);
    double spsi, cpsi;
    sincos(z[ZPSI], &spsi, &cpsi);
    if (part != 1 && part != 5) {
        for (int j = (part == 4 ? 0 : ell_first()); j < d.n_slk; j += (part == 4 ? 1 : ell_step)) {
            slk_row_eval(d, z, p + own_delta, pstride, j, off, spsi, cpsi, slack, ro);
            W[ZPSI][ZPSI] += lamh(d.n_lin + j) * ro.Hpp;          // the row is linear in (x, y); psi enters through the disc offset
            sink(d.n_lin + j, ro);
        }
    }
    if (part != 4) {
        // the two parts share the ellipsoid rows (part 1 the first half, next to the dynamics)
        const int j0 = part == 2 ? d.M / 2 : 0, j1 = part == 1 ? d.M / 2 : d.M;
        double risk_of = -1.0, ye = 0.0;                      // (Gaussian rows) the quantile of the last risk level seen: one level per configuration
        (void)risk_of; (void)ye;                              // in the reference (CONFIG probabilistic/risk), so it is evaluated once per stage, not once per row
        for (int j = j0 + ell_first(); j < j1; j += ell_step) {
            if constexpr (cm_gaussian_rows(CM)) {
                const double risk = TMPC_LDP(p + (size_t)ip_gauss(d, j, 4) * pstride);
                if (risk != risk_of) { ye = gauss_quantile(risk); risk_of = risk; }
                gauss_row_eval(d, z, p, pstride, j, r_disc, off, spsi, cpsi, ye, ro);
            } else {
                ellipsoid_row_eval(d, z, p, pstride, j, r_disc, off, spsi, cpsi, ro);
            }
            row_add_hessian(ro, lamh(d.n_up + j), W);
            sink(d.n_up + j, ro);
        }
    }
#endif
}

}  // namespace tmpc
