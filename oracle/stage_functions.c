/*
 * oracle/stage_functions.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Restates the NLP stage functions the reference defines symbolically in python and CasADi differentiates:
 * stage cost, inequality rows h, continuous dynamics and the acados ERK4 discretisation, each with exact
 * gradient and Hessian via second-order jets (oracle/jet.h).
 */
#include <stdlib.h>
#include "jet.h"
#include "tmpc_oracle.h"

/* z index map: inputs first, then states (solver_model.py:118-128, 200-201) */
enum { Z_A = 0, Z_W = 1, Z_X = 2, Z_Y = 3, Z_PSI = 4, Z_V = 5, Z_S = 6, Z_SLACK = 7 /* slack build only */ };

/* ---------------------------------------------------------------------------------------------
 * Parameter index map.  Rule (solver_definition.py:5-16, util/parameters.py:25-55): objective modules'
 * parameters first, then constraint modules', in module insertion order; duplicates skipped.
 * MPCBaseModule(a,w,v) -> acceleration, angular_velocity, velocity, reference_velocity
 * (generate_jackalsimulator_solver.py:45-52); ContouringObjective.define_parameters (contouring.py:22-46)
 * -> contour, lag, terminal_angle, terminal_contouring, then 9 per segment; LinearConstraints
 * (guidance_constraints.py:73-78) -> a1,a2,b per row; EllipsoidConstraint (ellipsoid_constraints.py:37-49)
 * -> ego_disc_radius, ego_disc_0_offset, then x,y,psi,major,minor,chi,r per obstacle.
 * Slack model (configuration_safe_horizon / rosnavigation configuration_tmpc,
 * generate_jackalsimulator_solver.py:70-93, generate_rosnavigation_solver.py:62-108): MPCBaseModule(a,w,slack,v)
 * -> acceleration, angular_velocity, slack, velocity, reference_velocity; Scenario / Decomp LinearConstraints
 * (scenario_constraints.py:40-49, decomp_constraints.py:44-52) -> ego_disc_0_offset (skipped when the ellipsoid
 * module already defined it), then a1,a2,b per row, scenario rows before decomp rows.
 * --------------------------------------------------------------------------------------------- */
static int w0(const orc_problem *pb) { return pb->cost_model == 2 ? 7 : 8 + pb->slack; }     /* number of weight parameters (goal stack: 4 base + goal_weight, goal_x, goal_y) */
int orc_idx_weight(const orc_problem *pb, int which)
{
    if (which == 8) return 2;                                       /* slack weight (slack build) */
    return which >= 2 ? which + pb->slack : which;
}
int orc_idx_spline(const orc_problem *pb, int seg, int which) { return w0(pb) + 9 * seg + which; }
int orc_idx_lin(const orc_problem *pb, int j, int which) { return w0(pb) + 9 * pb->S + 3 * j + which; }
int orc_idx_disc_radius(const orc_problem *pb) { return w0(pb) + 9 * pb->S + 3 * pb->n_lin; }
int orc_idx_disc_offset(const orc_problem *pb) { return orc_idx_disc_radius(pb) + (pb->M > 0 ? 1 : 0); }
int orc_idx_ellipsoid(const orc_problem *pb, int j, int which) { return orc_idx_disc_radius(pb) + 2 + 7 * j + which; }
int orc_idx_slk(const orc_problem *pb, int j, int which)
{
    const int base = pb->M > 0 ? orc_idx_ellipsoid(pb, pb->M, 0) : orc_idx_disc_radius(pb) + 1;
    return base + 3 * j + which;
}
int orc_model_nx(void) { return ORC_NXE; }
/* GaussianConstraint.define_parameters (gaussian_constraints.py:40-52): ego_disc_radius, ego_disc_0_offset, then
 * x, y, major, minor, risk, r per obstacle */
int orc_idx_gaussian(const orc_problem *pb, int j, int which) { return orc_idx_disc_radius(pb) + 2 + 6 * j + which; }
void orc_problem_set_gaussian(orc_problem *pb, int n_gauss)
{
    if (pb->M != 0 || pb->n_slk != 0) abort();
    pb->n_gauss = n_gauss;
    pb->npar = 8 + ORC_SLACK + 9 * pb->S + 3 * pb->n_lin + (n_gauss > 0 ? 2 + 6 * n_gauss : 0);
}

void orc_problem_set_goal_stack(orc_problem *pb, int model)
{
    if (pb->slack || pb->n_slk != 0 || pb->n_gauss != 0) abort();
    pb->cost_model = 2; pb->model = model; pb->S = 0; pb->n_lin = 0;
    pb->npar = 7 + (pb->M > 0 ? 2 + 7 * pb->M : 0);
    if (model == 1) {        /* solver_model.py:179-180 SecondOrderUnicycleModel bounds [a, w, x, y, psi, v]; the padding slot keeps the spline's */
        const double lb[6] = {-2.0, -2.0, -200.0, -200.0, -M_PI * 4, -2.0}, ub[6] = {2.0, 2.0, 200.0, 200.0, M_PI * 4, 3.0};
        for (int i = 0; i < 6; i++) { pb->lb[i] = lb[i]; pb->ub[i] = ub[i]; }
    }
}

void orc_problem_init(orc_problem *pb, int N, int S, int n_lin, int M) { orc_problem_init_ex(pb, N, S, n_lin, M, 0); }

void orc_problem_set_hpipm_like(orc_problem *pb, int warm_start)
{
    pb->ipm_mu0 = 10.0; pb->ipm_thr0 = 0.1; pb->ipm_tau = 0.995; pb->ipm_init_box = 1;
    pb->qp_warm_start = warm_start;
}

void orc_problem_init_ex(orc_problem *pb, int N, int S, int n_lin, int M, int n_slk)
{
    pb->N = N; pb->S = S; pb->n_lin = n_lin; pb->M = M; pb->n_slk = n_slk; pb->slack = ORC_SLACK; pb->n_gauss = 0;
    pb->npar = 8 + ORC_SLACK + 9 * S + 3 * n_lin + (M > 0 ? 2 + 7 * M : 0) + (n_slk > 0 ? (M > 0 ? 0 : 1) + 3 * n_slk : 0);
    pb->lb_slack = 0.0; pb->ub_slack = 5000.0;     /* solver_model.py:285-286 */
    pb->dt = 0.2;                 /* settings.yaml:3 integrator_step */
    pb->n_sqp = 10;               /* settings.yaml:16 */
    pb->qp_iter_max = 50;         /* generate_acados_solver.py:172 */
    pb->qp_tol = 1e-5;            /* generate_acados_solver.py:162 */
    pb->reg_eps = 1e-4;           /* [UPSTREAM] acados reg_epsilon default */
    pb->ipm_mu0 = 1e-2;           /* own IPM (any converged QP solver reproduces the unique QP solution); tuned: fewest iterations */
    pb->ipm_thr0 = 1e-2;
    pb->ipm_tau = 0.999;
    pb->erk_steps = 3;            /* generate_acados_solver.py:150 */
    pb->cost_model = 0;
    pb->qp_warm_start = 0; pb->ipm_init_box = 0; pb->riccati_form = 0; pb->model = 0;
    /* solver_model.py:204-205 ContouringSecondOrderUnicycleModel bounds, order [a,w,x,y,psi,v,spline] */
    const double lb[ORC_NV] = {-2.0, -0.8, -2000.0, -2000.0, -M_PI * 4, -0.01, -1.0};
    const double ub[ORC_NV] = {2.0, 0.8, 2000.0, 2000.0, M_PI * 4, 3.0, 10000.0};
    for (int i = 0; i < ORC_NV; i++) { pb->lb[i] = lb[i]; pb->ub[i] = ub[i]; }
}

/* ---------------------------------------------------------------------------------------------
 * Glued cubic spline (solver_generator/spline.py).
 * --------------------------------------------------------------------------------------------- */
typedef struct { double a, b, c, d, start; } seg_t;

/* SplineSegment.at (spline.py:16-18) */
static jet seg_at(const seg_t *sg, jet s)
{
    jet t = jet_addc(s, -sg->start);
    jet t2 = jet_mul(t, t), t3 = jet_mul(t2, t);
    jet r = jet_scale(t3, sg->a);
    r = jet_add(r, jet_scale(t2, sg->b));
    r = jet_add(r, jet_scale(t, sg->c));
    return jet_addc(r, sg->d);
}
/* SplineSegment.deriv (spline.py:20-22) */
static jet seg_deriv(const seg_t *sg, jet s)
{
    jet t = jet_addc(s, -sg->start);
    jet r = jet_scale(jet_mul(t, t), 3.0 * sg->a);
    r = jet_add(r, jet_scale(t, 2.0 * sg->b));
    return jet_addc(r, sg->c);
}
/* SplineSegment.deriv2 (spline.py:24-26) */
static jet seg_deriv2(const seg_t *sg, jet s)
{
    jet t = jet_addc(s, -sg->start);
    return jet_addc(jet_scale(t, 6.0 * sg->a), 2.0 * sg->b);
}
/* Spline.__init__ lambdas (spline.py:37): 1/(1+exp((s - start_i + 0.02)/0.1)), i = 1..S-1 */
static jet glue_lambda(double start_i, jet s)
{
    jet e = jet_exp(jet_scale(jet_addc(s, -start_i + 0.02), 1.0 / 0.1));
    return jet_recip(jet_addc(e, 1.0));
}
/* Spline.at / Spline.deriv (spline.py:39-50): the SAME blend is applied to values and to segment
 * derivatives (deriv is not d/ds of at). */
/* deriv = 2: Spline.deriv2 (spline.py:52-56), again the same blend, on the segments' second derivatives */
static jet spline_blend(const seg_t *sg, const jet *lam, int S, jet s, int deriv)
{
    jet value = deriv == 2 ? seg_deriv2(&sg[S - 1], s) : deriv ? seg_deriv(&sg[S - 1], s) : seg_at(&sg[S - 1], s);
    for (int k = S - 1; k >= 1; k--) {
        jet prev = deriv == 2 ? seg_deriv2(&sg[k - 1], s) : deriv ? seg_deriv(&sg[k - 1], s) : seg_at(&sg[k - 1], s);
        jet one_minus = jet_addc(jet_neg(lam[k - 1]), 1.0);
        value = jet_add(jet_mul(lam[k - 1], prev), jet_mul(one_minus, value));
    }
    return value;
}

static void load_segments(const orc_problem *pb, const double *p, seg_t *sx, seg_t *sy)
{
    for (int i = 0; i < pb->S; i++) {
        sx[i].a = p[orc_idx_spline(pb, i, 0)]; sx[i].b = p[orc_idx_spline(pb, i, 1)];
        sx[i].c = p[orc_idx_spline(pb, i, 2)]; sx[i].d = p[orc_idx_spline(pb, i, 3)];
        sy[i].a = p[orc_idx_spline(pb, i, 4)]; sy[i].b = p[orc_idx_spline(pb, i, 5)];
        sy[i].c = p[orc_idx_spline(pb, i, 6)]; sy[i].d = p[orc_idx_spline(pb, i, 7)];
        sx[i].start = sy[i].start = p[orc_idx_spline(pb, i, 8)];
    }
}

static void jet_out(const jet *j, double *val, double *grad, double *hess)
{
    if (val) *val = j->v;
    if (grad) for (int i = 0; i < ORC_NVE; i++) grad[i] = j->g[i];
    if (hess) for (int i = 0; i < ORC_NVE; i++) for (int k = 0; k < ORC_NVE; k++) hess[i * ORC_NVE + k] = j->H[i][k];
}

/* ---------------------------------------------------------------------------------------------
 * Stage cost at stage_idx = 1 (generate_acados_solver.py:48): no terminal terms under acados
 * (contouring.py:84-96 inactive).  NOT scaled by dt here.
 *   MPCBaseModule / WeightsObjective.get_value (mpc_base.py:47-60) with the weigh_variable calls of
 *   generate_jackalsimulator_solver.py:45-52:  w_a a^2 + w_w w^2 + w_v (v - v_ref)^2
 *   ContouringObjective.get_value (contouring.py:48-98): w_lag e_l^2 + w_contour e_c^2
 * --------------------------------------------------------------------------------------------- */
void orc_stage_cost(const orc_problem *pb, const double *z, const double *p,
                    double *val, double grad[ORC_NVE], double hess[ORC_NVE * ORC_NVE])
{
    jet a = jet_var(z[Z_A], Z_A), w = jet_var(z[Z_W], Z_W), x = jet_var(z[Z_X], Z_X), y = jet_var(z[Z_Y], Z_Y);
    jet v = jet_var(z[Z_V], Z_V), s = jet_var(z[Z_S], Z_S);
    const double w_a = p[orc_idx_weight(pb, 0)], w_w = p[orc_idx_weight(pb, 1)], w_v = p[orc_idx_weight(pb, 2)];
    const double v_ref = p[orc_idx_weight(pb, 3)], w_contour = p[orc_idx_weight(pb, 4)], w_lag = p[orc_idx_weight(pb, 5)];

    /* objectives are summed in module order (solver_definition.py:26-28), starting from cost = 0.0 */
    jet cost = jet_const(0.0);
    cost = jet_add(cost, jet_scale(jet_sq(a), w_a));                  /* w[0] * x**2 */
    cost = jet_add(cost, jet_scale(jet_sq(w), w_w));
#if ORC_SLACK
    {   /* weigh_variable("slack", "slack") sits between w and v (generate_jackalsimulator_solver.py:79-86) */
        jet sl = jet_var(z[Z_SLACK], Z_SLACK);
        cost = jet_add(cost, jet_scale(jet_sq(sl), p[orc_idx_weight(pb, 8)]));
    }
#endif
    cost = jet_add(cost, jet_scale(jet_sq(jet_addc(v, -v_ref)), w_v));/* w[0] * (x - w[1])**2 */

    if (pb->cost_model == 2) {
        /* GoalObjective.get_value (goal_module.py:22-36): goal_weight ((x - goal_x)^2 + (y - goal_y)^2) / (goal_x^2 + goal_y^2 + 0.01) */
        const double gw = p[4], gx = p[5], gy = p[6];
        jet d2 = jet_add(jet_sq(jet_addc(x, -gx)), jet_sq(jet_addc(y, -gy)));
        cost = jet_add(cost, jet_scale(jet_scale(d2, gw), 1.0 / (gx * gx + gy * gy + 0.01)));
        jet_out(&cost, val, grad, hess);
        return;
    }
    seg_t sx[16], sy[16]; jet lam[16];
    load_segments(pb, p, sx, sy);
    for (int i = 1; i < pb->S; i++) lam[i - 1] = glue_lambda(sx[i].start, s);
    jet path_x = spline_blend(sx, lam, pb->S, s, 0);                  /* contouring.py:69-70 */
    jet path_y = spline_blend(sy, lam, pb->S, s, 0);
    jet dx = spline_blend(sx, lam, pb->S, s, 1);                      /* deriv_normalized, spline.py:72-77 */
    jet dy = spline_blend(sy, lam, pb->S, s, 1);
    jet norm = jet_sqrt(jet_add(jet_mul(dx, dx), jet_mul(dy, dy)));
    jet dxn = jet_div(dx, norm), dyn = jet_div(dy, norm);

    jet ex = jet_sub(x, path_x), ey = jet_sub(y, path_y);
    if (pb->cost_model == 1) {
        /* CurvatureAwareContouringObjective.get_value (curvature_aware_contouring.py:48-105) at stage_idx = 1:
         *   projection_ratio = 1 / (1 - ((x - X) X'' + (y - Y) Y''))                                  (:82-83, path.deriv2)
         *   s_dot = v (cos psi X'n + sin psi Y'n) projection_ratio                                     (:84)
         *   cost += contour ((x - X)^2 + (y - Y)^2) + velocity (s_dot - reference_velocity)^2         (:87-90)
         * (`lag` is a parameter of the module but not used, :26; the terminal terms :93-105 are Forces-only like contouring.py:84-96) */
        jet psi = jet_var(z[Z_PSI], Z_PSI);
        jet ddx = spline_blend(sx, lam, pb->S, s, 2), ddy = spline_blend(sy, lam, pb->S, s, 2);
        jet ratio = jet_recip(jet_addc(jet_neg(jet_add(jet_mul(ex, ddx), jet_mul(ey, ddy))), 1.0));
        jet s_dot = jet_mul(jet_mul(v, jet_add(jet_mul(jet_cos(psi), dxn), jet_mul(jet_sin(psi), dyn))), ratio);
        jet dist2 = jet_add(jet_sq(ex), jet_sq(ey));
        cost = jet_add(cost, jet_scale(dist2, w_contour));
        cost = jet_add(cost, jet_scale(jet_sq(jet_addc(s_dot, -v_ref)), w_v));
        jet_out(&cost, val, grad, hess);
        return;
    }
    jet contour_error = jet_sub(jet_mul(dyn, ex), jet_mul(dxn, ey));  /* contouring.py:74 */
    jet lag_error = jet_add(jet_mul(dxn, ex), jet_mul(dyn, ey));      /* contouring.py:75 */
    cost = jet_add(cost, jet_scale(jet_sq(lag_error), w_lag));        /* :77 */
    cost = jet_add(cost, jet_scale(jet_sq(contour_error), w_contour));/* :78 */
    jet_out(&cost, val, grad, hess);
}

/* ---------------------------------------------------------------------------------------------
 * Inequality rows, order = module/constraint order (solver_definition.py:37-49):
 *   LinearConstraints.get_constraints (guidance_constraints.py:95-110):  a1*x + a2*y - b   (<= 0)
 *   EllipsoidConstraint.get_constraints (ellipsoid_constraints.py:65-119):
 *       d^T R(psi_o)^T diag(1/(maj*sqrt(chi)+r_disc+r)^2, 1/(min*sqrt(chi)+r_disc+r)^2) R(psi_o) d  (>= 1)
 *       d = pos + R(psi) [offset, 0] - obstacle,   R = rotation_matrix (util/math.py:5-7)
 * --------------------------------------------------------------------------------------------- */
void orc_stage_constraints(const orc_problem *pb, const double *z, const double *p,
                           double *h, double *jac, double *hess)
{
    jet x = jet_var(z[Z_X], Z_X), y = jet_var(z[Z_Y], Z_Y), psi = jet_var(z[Z_PSI], Z_PSI);
    int row = 0;
    for (int j = 0; j < pb->n_lin; j++, row++) {
        double a1 = p[orc_idx_lin(pb, j, 0)], a2 = p[orc_idx_lin(pb, j, 1)], b = p[orc_idx_lin(pb, j, 2)];
        jet c = jet_addc(jet_add(jet_scale(x, a1), jet_scale(y, a2)), -b);
        jet_out(&c, &h[row], jac ? &jac[row * ORC_NVE] : 0, hess ? &hess[row * ORC_NVE * ORC_NVE] : 0);
    }
    if (pb->M == 0 && pb->n_slk == 0 && pb->n_gauss == 0) return;
    const double r_disc = (pb->M > 0 || pb->n_gauss > 0) ? p[orc_idx_disc_radius(pb)] : 0.0;
    const double disc_x = p[orc_idx_disc_radius(pb) + ((pb->M > 0 || pb->n_gauss > 0) ? 1 : 0)];
    /* disc_pos = pos + rotation_car @ [disc_x, 0]  (ellipsoid_constraints.py:109-111) */
    jet dpx = jet_add(x, jet_scale(jet_cos(psi), disc_x));
    jet dpy = jet_add(y, jet_scale(jet_sin(psi), disc_x));
    for (int j = 0; j < pb->M; j++, row++) {
        double ox = p[orc_idx_ellipsoid(pb, j, 0)], oy = p[orc_idx_ellipsoid(pb, j, 1)];
        double opsi = p[orc_idx_ellipsoid(pb, j, 2)];
        double major = p[orc_idx_ellipsoid(pb, j, 3)], minor = p[orc_idx_ellipsoid(pb, j, 4)];
        double chi = p[orc_idx_ellipsoid(pb, j, 5)], r = p[orc_idx_ellipsoid(pb, j, 6)];
        major *= sqrt(chi); minor *= sqrt(chi);                                  /* :94-95 */
        double ab00 = 1.0 / ((major + r_disc + r) * (major + r_disc + r));       /* :97 */
        double ab11 = 1.0 / ((minor + r_disc + r) * (minor + r_disc + r));       /* :100 */
        double c = cos(opsi), s = sin(opsi);
        /* R^T ab R with R = [[c,-s],[s,c]] */
        double m00 = c * c * ab00 + s * s * ab11;
        double m01 = -c * s * ab00 + s * c * ab11;
        double m11 = s * s * ab00 + c * c * ab11;
        jet d0 = jet_addc(dpx, -ox), d1 = jet_addc(dpy, -oy);
        jet q = jet_add(jet_add(jet_scale(jet_sq(d0), m00), jet_scale(jet_mul(d0, d1), 2.0 * m01)),
                        jet_scale(jet_sq(d1), m11));
        jet_out(&q, &h[row], jac ? &jac[row * ORC_NVE] : 0, hess ? &hess[row * ORC_NVE * ORC_NVE] : 0);
    }
    /* GaussianConstraint.get_constraints (gaussian_constraints.py:66-113):
     *   a = diff/|diff|,  a.diff - (r_disc + r) - erfinv(1 - 2 risk) * sqrt(2 a^T diag(major^2, minor^2) a)   (>= 0)
     * erfinv by the rational start + two Newton steps of :103-111; it depends on the parameters only. */
    for (int j = 0; j < pb->n_gauss; j++, row++) {
        double ox = p[orc_idx_gaussian(pb, j, 0)], oy = p[orc_idx_gaussian(pb, j, 1)];
        double sx = p[orc_idx_gaussian(pb, j, 2)], sy = p[orc_idx_gaussian(pb, j, 3)];
        double risk = p[orc_idx_gaussian(pb, j, 4)], r = p[orc_idx_gaussian(pb, j, 5)];
        double xe = 1.0 - 2.0 * risk;
        double zz = sqrt(-log((1.0 - xe) / 2.0));
        double ye = (((1.641345311 * zz + 3.429567803) * zz - 1.624906493) * zz - 1.970840454) / ((1.637067800 * zz + 3.543889200) * zz + 1.0);
        for (int it = 0; it < 2; it++) ye = ye - (erf(ye) - xe) / (2.0 / sqrt(M_PI) * exp(-ye * ye));
        jet d0 = jet_addc(dpx, -ox), d1 = jet_addc(dpy, -oy);
        jet nrm = jet_sqrt(jet_add(jet_sq(d0), jet_sq(d1)));
        jet a0 = jet_div(d0, nrm), a1 = jet_div(d1, nrm);
        jet along = jet_add(jet_mul(a0, d0), jet_mul(a1, d1));                       /* a_ij.T @ diff_pos */
        jet quad = jet_add(jet_scale(jet_sq(a0), sx * sx), jet_scale(jet_sq(a1), sy * sy));
        jet c = jet_sub(jet_addc(along, -(r_disc + r)), jet_scale(jet_sqrt(jet_scale(quad, 2.0)), ye));
        jet_out(&c, &h[row], jac ? &jac[row * ORC_NVE] : 0, hess ? &hess[row * ORC_NVE * ORC_NVE] : 0);
    }
    /* Scenario / Decomp LinearConstraints.get_constraints (scenario_constraints.py:64-94, decomp_constraints.py:68-98):
     * a1*disc_pos[0] + a2*disc_pos[1] - (b + slack); slack = 0.0 when the model has no slack state
     * (decomp_constraints.py:78-84) */
    for (int j = 0; j < pb->n_slk; j++, row++) {
        double a1 = p[orc_idx_slk(pb, j, 0)], a2 = p[orc_idx_slk(pb, j, 1)], b = p[orc_idx_slk(pb, j, 2)];
        jet c = jet_add(jet_scale(dpx, a1), jet_scale(dpy, a2));
#if ORC_SLACK
        c = jet_sub(c, jet_addc(jet_var(z[Z_SLACK], Z_SLACK), b));
#else
        c = jet_addc(c, -b);
#endif
        jet_out(&c, &h[row], jac ? &jac[row * ORC_NVE] : 0, hess ? &hess[row * ORC_NVE * ORC_NVE] : 0);
    }
}

/* bounds: lin rows (-inf, 0] (guidance_constraints.py:83-93); ellipsoid rows [1, +inf)
 * (ellipsoid_constraints.py:51-63); +-inf -> +-1e15 (generate_acados_solver.py:17-24) */
void orc_constraint_bounds(const orc_problem *pb, double *lh, double *uh)
{
    int row = 0;
    for (int j = 0; j < pb->n_lin; j++, row++) { lh[row] = -1e15; uh[row] = 0.0; }
    for (int j = 0; j < pb->M; j++, row++) { lh[row] = 1.0; uh[row] = 1e15; }
    for (int j = 0; j < pb->n_gauss; j++, row++) { lh[row] = 0.0; uh[row] = 1e15; }   /* gaussian_constraints.py:54-64 */
    for (int j = 0; j < pb->n_slk; j++, row++) { lh[row] = -1e15; uh[row] = 0.0; }   /* scenario_constraints.py:52-62 */
}

/* ---------------------------------------------------------------------------------------------
 * Dynamics: ContouringSecondOrderUnicycleModel.continuous_model (solver_model.py:207-214)
 *   xdot = [v cos psi, v sin psi, w, a, v]
 * --------------------------------------------------------------------------------------------- */
void orc_continuous_dynamics(const double *z, double f[ORC_NXE])
{
    f[0] = z[Z_V] * cos(z[Z_PSI]); f[1] = z[Z_V] * sin(z[Z_PSI]); f[2] = z[Z_W]; f[3] = z[Z_A]; f[4] = z[Z_V];
#if ORC_SLACK
    f[5] = 0.0;                  /* solver_model.py:288-296 */
#endif
}

static void f_jet(const jet *x /* NX */, const jet *u /* NU */, jet *f, int model)
{
    f[0] = jet_mul(x[3], jet_cos(x[2]));
    f[1] = jet_mul(x[3], jet_sin(x[2]));
    f[2] = u[1];
    f[3] = u[0];
    f[4] = model == 1 ? jet_const(0.0) : x[3];        /* SecondOrderUnicycleModel (solver_model.py:183-191) has no fifth state: the padding slot stands still */
#if ORC_SLACK
    f[5] = jet_const(0.0);
#endif
}

/* acados ERK integrator: classic 4-stage RK4, sim_method_num_steps = 3 sub-steps per shooting
 * interval of length dt (generate_acados_solver.py:143,148-150); controls constant over the interval. */
void orc_discrete_dynamics(const orc_problem *pb, const double *z, double xnext[ORC_NXE], double *jac, double *hess)
{
    jet u[ORC_NU], x[ORC_NXE], k1[ORC_NXE], k2[ORC_NXE], k3[ORC_NXE], k4[ORC_NXE], xt[ORC_NXE];
    for (int i = 0; i < ORC_NU; i++) u[i] = jet_var(z[i], i);
    for (int i = 0; i < ORC_NXE; i++) x[i] = jet_var(z[ORC_NU + i], ORC_NU + i);
    const double h = pb->dt / pb->erk_steps;
    for (int step = 0; step < pb->erk_steps; step++) {
        f_jet(x, u, k1, pb->model);
        for (int i = 0; i < ORC_NXE; i++) xt[i] = jet_add(x[i], jet_scale(k1[i], 0.5 * h));
        f_jet(xt, u, k2, pb->model);
        for (int i = 0; i < ORC_NXE; i++) xt[i] = jet_add(x[i], jet_scale(k2[i], 0.5 * h));
        f_jet(xt, u, k3, pb->model);
        for (int i = 0; i < ORC_NXE; i++) xt[i] = jet_add(x[i], jet_scale(k3[i], h));
        f_jet(xt, u, k4, pb->model);
        for (int i = 0; i < ORC_NXE; i++) {
            jet sum = jet_add(jet_add(k1[i], jet_scale(k2[i], 2.0)), jet_add(jet_scale(k3[i], 2.0), k4[i]));
            x[i] = jet_add(x[i], jet_scale(sum, h / 6.0));
        }
    }
    for (int i = 0; i < ORC_NXE; i++)
        jet_out(&x[i], &xnext[i], jac ? &jac[i * ORC_NVE] : 0, hess ? &hess[i * ORC_NVE * ORC_NVE] : 0);
}
