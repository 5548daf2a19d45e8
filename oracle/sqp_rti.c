/*
 * oracle/sqp_rti.c -- CPU ORACLE (test infrastructure).
 *
 * Restates Solver::solve() (mpc_planner_solver/src/acados_solver_interface.cpp:86-204) for the acados
 * configuration of solver_generator/generate_acados_solver.py:143-177:
 *   SQP_RTI, EXACT Hessian, MIRROR, FIXED_STEP (full step), ERK4 x 3, QP to qp_tol, <= 50 QP iterations.
 * [UPSTREAM] assumptions (acados absent from /root/reference), all listed in DESIGN.md:
 *   U1 stage costs k<N are scaled by the shooting interval dt; terminal cost is zero (cost_type_e unset),
 *   U2 h(z,p) applies at nodes 0..N-1 (incl. node 0, hence the k=0 dummies written by the C++ modules),
 *      box bounds on u at 0..N-1, on x at 1..N-1, none at node N; x_0 = xinit,
 *   U3 bounds with |value| >= 1e10 are treated as absent (one-sided rows),
 *   U4 MIRROR eps = 1e-4; W_N = 0 -> eps*I,
 *   U5 a QP that stops at its iteration limit still takes the step (status stays success, qp_status=2);
 *      NaN / min-step QP failures return ACADOS_QP_FAILURE (4) without taking the step,
 *   U6 res_eq = inf-norm of the dynamics defects + initial-condition violation at the returned iterate.
 *   U9 (slack build) the slack state obeys slack_{k+1} = slack_k and x_0 = xinit fixes it at node 0
 *      (ocp.constraints.x0 covers all nx states, generate_acados_solver.py:95), so every QP determines its step
 *      without any freedom: d_slack_0 = xinit_s - slack_0, d_slack_{k+1} = d_slack_k + (slack_k - slack_{k+1}).
 *      The QP is strictly convex after MIRROR, hence its primal solution is unique and equals the solution of the
 *      QP with that chain substituted (QP presolve): the substituted QP has the 7 remaining variables per node,
 *      the slack column of the rows moves into their right-hand sides, and the slack box rows (0 <= slack <= 5000,
 *      now constants) drop out.  HPIPM would carry the chain as equalities; a converged solve gives the same step.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "qp.h"
#include "tmpc_oracle.h"

#define INF_BOUND 1e10

typedef struct {
    double z[ORC_MAX_N + 1][ORC_NVE];      /* iterate, [u_k; x_k]; node N: u unused */
    double pi[ORC_MAX_N + 1][ORC_NX];      /* dynamics multipliers (index = node of x_{k}) */
    double lam_h[ORC_MAX_N][ORC_MAX_NH];   /* (lam_upper - lam_lower) per general row */
} nlp_state;

static void build_qp(const orc_problem *pb, const nlp_state *st, const double *xinit, const double *params,
                     orc_qp *qp, int (*row_lo)[ORC_MAX_NH], int (*row_hi)[ORC_MAX_NH], orc_debug *dbg, double *dslack)
{
    const int N = pb->N, nh = pb->n_lin + pb->M + pb->n_gauss + pb->n_slk;
    const int E = ORC_NVE;                                   /* stride of the model's derivative arrays */
#if ORC_SLACK
    dslack[0] = xinit[ORC_NX] - st->z[0][ORC_NV];            /* U9 */
    for (int k = 0; k < N; k++) dslack[k + 1] = dslack[k] + (st->z[k][ORC_NV] - st->z[k + 1][ORC_NV]);
#else
    for (int k = 0; k <= N; k++) dslack[k] = 0.0;
#endif
    double lh[ORC_MAX_NH], uh[ORC_MAX_NH];
    orc_constraint_bounds(pb, lh, uh);
    qp->N = N;
    for (int k = 0; k < N; k++) {
        const double *p = &params[(size_t)k * pb->npar];     /* update_params(k, all_parameters[k*NP]) :127-135 */
        const double *z = st->z[k];
        double xn[ORC_NXE], A[ORC_NXE * ORC_NVE], Hd[ORC_NXE * ORC_NVE * ORC_NVE];
        orc_discrete_dynamics(pb, z, xn, A, Hd);
        double l, gl[ORC_NVE], Hl[ORC_NVE * ORC_NVE];
        orc_stage_cost(pb, z, p, &l, gl, Hl);
        double h[ORC_MAX_NH], D[ORC_MAX_NH * ORC_NVE];
        static __thread double *tls_Hh = NULL;                 /* per-thread scratch, reused (see qp_ipm.c) */
        if (!tls_Hh) tls_Hh = (double *)malloc(sizeof(double) * ORC_MAX_NH * ORC_NVE * ORC_NVE);
        double *Hh = tls_Hh;
        orc_stage_constraints(pb, z, p, h, D, Hh);

        /* Lagrangian Hessian on the QP's 7 variables (the slack row/column of the model's Hessian is diagonal:
         * no curvature couples slack to the other variables, so MIRROR acts on the two blocks independently) */
        double W[ORC_NV * ORC_NV];
        for (int a = 0; a < ORC_NV; a++)
            for (int c = 0; c < ORC_NV; c++) {
                double acc = pb->dt * Hl[a * E + c];                                          /* U1 */
                for (int j = 0; j < ORC_NX; j++) acc += st->pi[k + 1][j] * Hd[j * E * E + a * E + c];
                for (int r = 0; r < nh; r++) acc += st->lam_h[k][r] * Hh[r * E * E + a * E + c];
                W[a * ORC_NV + c] = acc;
            }
        if (dbg) {
            memcpy(&dbg->W_raw[k * ORC_NV * ORC_NV], W, sizeof W);
            for (int j = 0; j < ORC_NV; j++) dbg->z_in[k * ORC_NV + j] = z[j];
            for (int j = 0; j < ORC_NX; j++) dbg->pi_in[(k + 1) * ORC_NX + j] = st->pi[k + 1][j];
            for (int r = 0; r < nh; r++) dbg->lamh_in[k * ORC_MAX_NH + r] = st->lam_h[k][r];
        }
        orc_mirror(W, ORC_NV, pb->reg_eps);
        for (int i = 0; i < ORC_NV; i++) {
            for (int j = 0; j < ORC_NV; j++) qp->W[k][i][j] = W[i * ORC_NV + j];
            qp->g[k][i] = pb->dt * gl[i];
        }
        for (int i = 0; i < ORC_NX; i++) {
            for (int j = 0; j < ORC_NV; j++) qp->BA[k][i][j] = A[i * E + j];
            qp->b[k][i] = xn[i] - st->z[k + 1][ORC_NU + i];
        }
        /* rows */
        int nr = 0;
        for (int r = 0; r < nh; r++) {
            row_lo[k][r] = row_hi[k][r] = -1;
            const double hs = h[r] + (ORC_SLACK ? D[r * E + ORC_NV] * dslack[k] : 0.0);   /* U9: slack column -> rhs */
            if (lh[r] > -INF_BOUND) {          /* lower: + (D dz - (lh - h)) >= 0   (U3) */
                for (int j = 0; j < ORC_NV; j++) qp->C[k][nr][j] = D[r * E + j];
                qp->sgn[k][nr] = 1.0; qp->beta[k][nr] = lh[r] - hs; row_lo[k][r] = nr; nr++;
            }
            if (uh[r] < INF_BOUND) {           /* upper: - (D dz - (uh - h)) >= 0 */
                for (int j = 0; j < ORC_NV; j++) qp->C[k][nr][j] = D[r * E + j];
                qp->sgn[k][nr] = -1.0; qp->beta[k][nr] = uh[r] - hs; row_hi[k][r] = nr; nr++;
            }
        }
        const int last_box = (k == 0) ? ORC_NU : ORC_NV;        /* U2: x_0 is fixed, not boxed */
        for (int j = 0; j < last_box; j++) {
            for (int side = 0; side < 2; side++) {
                for (int c = 0; c < ORC_NV; c++) qp->C[k][nr][c] = (c == j) ? 1.0 : 0.0;
                qp->sgn[k][nr] = side == 0 ? 1.0 : -1.0;
                qp->beta[k][nr] = (side == 0 ? pb->lb[j] : pb->ub[j]) - z[j];
                nr++;
            }
        }
        qp->nrow[k] = nr;
        if (dbg) {
            memcpy(&dbg->W[k * ORC_NV * ORC_NV], W, sizeof W);
            for (int i = 0; i < ORC_NV; i++) dbg->g[k * ORC_NV + i] = qp->g[k][i];
            for (int i = 0; i < ORC_NX; i++) for (int j = 0; j < ORC_NV; j++) dbg->BA[(k * ORC_NX + i) * ORC_NV + j] = A[i * E + j];
            for (int i = 0; i < ORC_NX; i++) dbg->b[k * ORC_NX + i] = qp->b[k][i];
            for (int r = 0; r < nh; r++) {
                dbg->h[k * ORC_MAX_NH + r] = h[r];
                for (int j = 0; j < ORC_NV; j++) dbg->D[(k * ORC_MAX_NH + r) * ORC_NV + j] = D[r * E + j];
            }
        }
    }
    /* terminal node: no cost (cost_type_e unset, generate_acados_solver.py:92), no constraints: W_N = eps I (U4) */
    memset(qp->W[N], 0, sizeof qp->W[N]); memset(qp->g[N], 0, sizeof qp->g[N]);
    {
        double WN[ORC_NX * ORC_NX]; memset(WN, 0, sizeof WN);
        orc_mirror(WN, ORC_NX, pb->reg_eps);
        for (int i = 0; i < ORC_NX; i++) for (int j = 0; j < ORC_NX; j++) qp->W[N][ORC_NU + i][ORC_NU + j] = WN[i * ORC_NX + j];
        if (dbg) {
            memset(&dbg->W[N * ORC_NV * ORC_NV], 0, sizeof(double) * ORC_NV * ORC_NV);
            for (int i = 0; i < ORC_NV; i++) {
                for (int j = 0; j < ORC_NV; j++) dbg->W[N * ORC_NV * ORC_NV + i * ORC_NV + j] = qp->W[N][i][j];
                dbg->g[N * ORC_NV + i] = 0.0;
            }
        }
    }
    qp->nrow[N] = 0;
    for (int j = 0; j < ORC_NX; j++) qp->dx0[j] = xinit[j] - st->z[0][ORC_NU + j];      /* lbx_0 = ubx_0 = xinit :124-125 */
}

/* optional per-QP interior-point iteration trace of the calling thread (tests / workload statistics) */
static __thread int *tls_qp_iter_trace = NULL;
void orc_set_qp_iter_trace(int *per_sqp_iteration) { tls_qp_iter_trace = per_sqp_iteration; }

static void solve_impl(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
                       double *xtraj, double *utraj, orc_info *info, orc_debug *dbg, int capture_sqp_iter,
                       int n_iter, double *pi_io, double *lamh_io);

void orc_solve_debug(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
                     double *xtraj, double *utraj, orc_info *info, orc_debug *dbg, int capture_sqp_iter)
{
    solve_impl(pb, xinit, x0, params, xtraj, utraj, info, dbg, capture_sqp_iter, pb->n_sqp, 0, 0);
}

/* The reference's capsules keep their multipliers between solves (acados_solver_interface.cpp:67-77 copies parameters only,
 * :274-284 overwrites the primal iterate only) and reset them when a solve does not succeed (:187-191).
 * pi_io [(N+1) NX] and lamh_io [N ORC_MAX_NH] (lam_upper - lam_lower per general row) are read as the starting multipliers
 * and overwritten with the final ones (zeros if exit_code != 1); n_iter RTI iterations. */
void orc_solve_carry(const orc_problem *pb, const double *xinit, const double *x0, const double *params, int n_iter,
                     double *pi_io, double *lamh_io, double *xtraj, double *utraj, orc_info *info)
{
    solve_impl(pb, xinit, x0, params, xtraj, utraj, info, 0, -1, n_iter, pi_io, lamh_io);
}

static void solve_impl(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
                       double *xtraj, double *utraj, orc_info *info, orc_debug *dbg, int capture_sqp_iter,
                       int n_iter, double *pi_io, double *lamh_io)
{
    const int N = pb->N, nh = pb->n_lin + pb->M + pb->n_gauss + pb->n_slk;
    double dslack[ORC_MAX_N + 1];
    /* per-thread workspaces, zeroed per solve (fresh solver state) but allocated once */
    static __thread nlp_state *tls_st = NULL; static __thread orc_qp *tls_qp = NULL; static __thread orc_qp_sol *tls_sol = NULL;
    static __thread int (*tls_lo)[ORC_MAX_NH] = NULL; static __thread int (*tls_hi)[ORC_MAX_NH] = NULL;
    if (!tls_st) {
        tls_st = (nlp_state *)malloc(sizeof(nlp_state)); tls_qp = (orc_qp *)malloc(sizeof(orc_qp));
        tls_sol = (orc_qp_sol *)malloc(sizeof(orc_qp_sol));
        tls_lo = malloc(ORC_MAX_N * sizeof *tls_lo); tls_hi = malloc(ORC_MAX_N * sizeof *tls_hi);
    }
    nlp_state *st = tls_st; orc_qp *qp = tls_qp; orc_qp_sol *sol = tls_sol;
    int (*row_lo)[ORC_MAX_NH] = tls_lo; int (*row_hi)[ORC_MAX_NH] = tls_hi;
    memset(st, 0, sizeof *st); memset(qp, 0, sizeof *qp); memset(sol, 0, sizeof *sol);
    memset(row_lo, 0, ORC_MAX_N * sizeof *row_lo); memset(row_hi, 0, ORC_MAX_N * sizeof *row_hi);

    /* loadWarmstart (:274-284): x_k = x0[nvar*k + nu ..], u_k = x0[nvar*k ..], k<N; x_N */
    for (int k = 0; k <= N; k++)
        for (int j = 0; j < ORC_NVE; j++) st->z[k][j] = x0[k * ORC_NVE + j];
    st->z[N][0] = st->z[N][1] = 0.0;
    if (pi_io) {
        for (int k = 0; k <= N; k++) for (int j = 0; j < ORC_NX; j++) st->pi[k][j] = pi_io[k * ORC_NX + j];
        for (int k = 0; k < N; k++) for (int r = 0; r < nh; r++) st->lam_h[k][r] = lamh_io[k * ORC_MAX_NH + r];
    }

    int status = 0;           /* acados status of the last Solver_acados_solve */
    info->qp_status = 0; info->sqp_iter = 0; info->qp_iter_total = 0;
    for (int it = 0; it < n_iter; it++) {                           /* :99-117 */
        build_qp(pb, st, xinit, params, qp, row_lo, row_hi, (dbg && it == capture_sqp_iter) ? dbg : 0, dslack);
        /* the first QP of a solve is cold (the QP memory is reset with every `*solver = *_solver` and after a failure,
         * acados_solver_interface.cpp:70,190); later ones follow qp_warm_start (generate_acados_solver.py:173 sets 2) */
        orc_qp_solve_form(qp, sol, pb->qp_iter_max, pb->qp_tol, pb->ipm_mu0, pb->ipm_thr0, pb->ipm_tau,
                          it > 0 ? pb->qp_warm_start : 0, pb->ipm_init_box, pb->riccati_form);
        info->qp_status = sol->status; info->sqp_iter = it + 1; info->qp_iter_total += sol->iters;
        if (tls_qp_iter_trace) tls_qp_iter_trace[it] = sol->iters;
        if (dbg && it == capture_sqp_iter) {
            for (int k = 0; k <= N; k++) {
                for (int j = 0; j < ORC_NV; j++) dbg->dz[k * ORC_NV + j] = sol->v[k][j];
                for (int j = 0; j < ORC_NX; j++) dbg->pi[k * ORC_NX + j] = sol->pi[k][j];
            }
            dbg->qp_iters = sol->iters;
        }
        if (sol->status != 0 && sol->status != 2) { status = 4; break; }   /* U5: ACADOS_QP_FAILURE, no step */
        status = 0;
        /* full step (globalization FIXED_STEP :158) + multipliers from the QP */
        for (int k = 0; k <= N; k++) {
            for (int j = (k == N ? ORC_NU : 0); j < ORC_NV; j++) st->z[k][j] += sol->v[k][j];
#if ORC_SLACK
            st->z[k][ORC_NV] += dslack[k];                                     /* U9 */
#endif
            if (k >= 1) for (int j = 0; j < ORC_NX; j++) st->pi[k][j] = sol->pi[k][j];
        }
        for (int k = 0; k < N; k++)
            for (int r = 0; r < nh; r++) {
                double lu = 0.0;                                  /* Hessian weight (lam_upper - lam_lower) */
                if (row_hi[k][r] >= 0) lu += sol->lam[k][row_hi[k][r]];
                if (row_lo[k][r] >= 0) lu -= sol->lam[k][row_lo[k][r]];
                st->lam_h[k][r] = lu;
            }
        if (info->qp_status != 0) break;                                   /* :105-106 (local `status` is never 0) */
    }

    /* completeOneIteration (:162-204) */
    double pobj = 0.0, res_eq = 0.0;
    for (int k = 0; k < N; k++) {
        double l; orc_stage_cost(pb, st->z[k], &params[(size_t)k * pb->npar], &l, 0, 0);
        pobj += pb->dt * l;                                                /* ocp_nlp_eval_cost, U1 */
        double xn[ORC_NXE]; orc_discrete_dynamics(pb, st->z[k], xn, 0, 0);
        for (int i = 0; i < ORC_NXE; i++) { double d = fabs(xn[i] - st->z[k + 1][ORC_NU + i]); if (d > res_eq) res_eq = d; }
    }
    for (int i = 0; i < ORC_NXE; i++) { double d = fabs(st->z[0][ORC_NU + i] - xinit[i]); if (d > res_eq) res_eq = d; }
    for (int k = 0; k <= N; k++) for (int i = 0; i < ORC_NXE; i++) xtraj[k * ORC_NXE + i] = st->z[k][ORC_NU + i];
    for (int k = 0; k < N; k++) for (int i = 0; i < ORC_NU; i++) utraj[k * ORC_NU + i] = st->z[k][i];
    if (res_eq > 1e-2 && status == 0) status = 4;                          /* :177-181 */
    if (!isfinite(pobj)) status = 4;
    /* map to Forces codes (:197-201): 0 -> 1, 1 -> 0 */
    int exit_code = status == 0 ? 1 : (status == 1 ? 0 : status);
    info->pobj = pobj; info->res_eq = res_eq; info->exit_code = exit_code;
    if (pi_io) {
        const int keep = exit_code == 1;                                   /* Solver_acados_reset on failure (:187-191) */
        for (int k = 0; k <= N; k++) for (int j = 0; j < ORC_NX; j++) pi_io[k * ORC_NX + j] = keep ? st->pi[k][j] : 0.0;
        for (int k = 0; k < N; k++) for (int r = 0; r < nh; r++) lamh_io[k * ORC_MAX_NH + r] = keep ? st->lam_h[k][r] : 0.0;
    }
}

void orc_solve(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
               double *xtraj, double *utraj, orc_info *info)
{
    orc_solve_debug(pb, xinit, x0, params, xtraj, utraj, info, 0, -1);
}
