/* oracle/qp.h -- CPU ORACLE (test infrastructure): OCP-structured QP container for the IPM. */
#ifndef ORC_QP_H
#define ORC_QP_H
#include "tmpc_oracle.h"

#define ORC_MAX_ROWS (2 * ORC_MAX_NH + 2 * ORC_NV)

typedef struct {
    int N;
    /* min sum_k 1/2 v_k^T W_k v_k + g_k^T v_k,  v_k = [du_k; dx_k] (node N: dx only, u-block unused) */
    double W[ORC_MAX_N + 1][ORC_NV][ORC_NV];
    double g[ORC_MAX_N + 1][ORC_NV];
    /* dx_{k+1} = [B A] v_k + b_k */
    double BA[ORC_MAX_N][ORC_NX][ORC_NV];
    double b[ORC_MAX_N][ORC_NX];
    double dx0[ORC_NX];                  /* dx_0 = xinit - x_0 (lbx_0 = ubx_0 = xinit) */
    /* one-sided rows  sgn * (c^T v_k - beta) >= 0 */
    int nrow[ORC_MAX_N + 1];
    double C[ORC_MAX_N + 1][ORC_MAX_ROWS][ORC_NV];
    double sgn[ORC_MAX_N + 1][ORC_MAX_ROWS];
    double beta[ORC_MAX_N + 1][ORC_MAX_ROWS];
} orc_qp;

typedef struct {
    double v[ORC_MAX_N + 1][ORC_NV];
    double pi[ORC_MAX_N + 1][ORC_NX];    /* pi[k] = multiplier of the dynamics equation defining x_k, k>=1 */
    double lam[ORC_MAX_N + 1][ORC_MAX_ROWS];
    double t[ORC_MAX_N + 1][ORC_MAX_ROWS];
    int iters;
    int status;                          /* 0 ok, 2 max iter, 3 min step, 4 NaN */
} orc_qp_sol;

void orc_qp_solve(const orc_qp *qp, orc_qp_sol *sol, int iter_max, double tol, double mu0, double thr0, double tau);
/* warm: 0 cold start; 1 primal start from `sol` as it stands (the previous QP's solution); 2 primal and dual (pi, lam, t) from `sol`.
 * init_box: cold / primal start moved into the interior of the box rows by thr0 (HPIPM's d_ocp_qp_init_var). */
void orc_qp_solve_ex(const orc_qp *qp, orc_qp_sol *sol, int iter_max, double tol, double mu0, double thr0, double tau, int warm, int init_box);
/* riccati_form: orc_problem::riccati_form (0 square-root, 1 input-block elimination only) */
void orc_qp_solve_form(const orc_qp *qp, orc_qp_sol *sol, int iter_max, double tol, double mu0, double thr0, double tau, int warm, int init_box, int riccati_form);

#endif
