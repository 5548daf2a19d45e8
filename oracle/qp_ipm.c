/*
 * oracle/qp_ipm.c -- CPU ORACLE (test infrastructure).
 *
 * Primal-dual interior-point solve of the OCP-structured QP that one SQP_RTI iteration poses
 * (SURVEY Appendix B step 4).  The reference delegates this to HPIPM (qp_solver =
 * PARTIAL_CONDENSING_HPIPM, no condensing horizon set => stage-structured Riccati factorisation,
 * generate_acados_solver.py:171-173), whose source is not in /root/reference.  After MIRROR every W_k
 * is positive definite, so the QP solution is unique and any solver converged to qp_tol reproduces
 * it; this file is a textbook Mehrotra predictor-corrector with a Riccati factor/solve per iteration.
 *
 * Unknowns per node k: v_k = [du_k; dx_k] (node N: dx only), multipliers pi_k of the dynamics
 * equations, and (lam_i, t_i) >= 0 for every one-sided row  sgn_i (c_i^T v - beta_i) - t_i = 0.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qp.h"

typedef struct {
    double Hh[ORC_MAX_N + 1][ORC_NV][ORC_NV];   /* W + sum (lam/t) c c^T */
    double gh[ORC_MAX_N + 1][ORC_NV];
    double rb[ORC_MAX_N][ORC_NX];
    double rg[ORC_MAX_N + 1][ORC_NV];
    double rd[ORC_MAX_N + 1][ORC_MAX_ROWS];
    double q[ORC_MAX_N + 1][ORC_MAX_ROWS];
    /* Riccati */
    double Lp[ORC_MAX_N + 1][ORC_NX][ORC_NX];   /* P_k = Lp Lp^T */
    double P[ORC_MAX_N + 1][ORC_NX][ORC_NX];    /* riccati_form 1: P_k itself */
    int classical;
    double L[ORC_MAX_N][ORC_NV][ORC_NV];        /* chol of the stage matrix F_k */
    double p[ORC_MAX_N + 1][ORC_NX];
    double y[ORC_MAX_N][ORC_NU];
    /* steps */
    double dv[ORC_MAX_N + 1][ORC_NV];
    double dpi[ORC_MAX_N + 1][ORC_NX];
    double dlam[ORC_MAX_N + 1][ORC_MAX_ROWS];
    double dt[ORC_MAX_N + 1][ORC_MAX_ROWS];
    int bad;
} ipm_ws;

/*
 * Square-root (Cholesky) Riccati recursion -- the form HPIPM uses [UPSTREAM], robust when the barrier
 * terms lam/t become huge.  With P_{k+1} = Lp Lp^T:
 *     G = Lp^T [B A]            (5 x 7)
 *     F = Hh_k + G^T G          (7 x 7, ordering [u; x])
 *     F = L L^T,  L = [Luu 0; Lxu Lxx]   =>   P_k = Lxx Lxx^T  (no subtraction of large numbers)
 * The factor L of every stage is kept for the vector solves.
 */
/*
 * orc_problem::riccati_form = 1: the same elimination stopped after the two input columns -- F = Hh + [B A]^T P [B A], Cholesky of the uu
 * block only, P_k = F_xx - Lxu Lxu^T kept as it is instead of being re-factorised (HPIPM's square_root_alg = 0 [UPSTREAM]; what the HIP
 * kernels run, csrc/tmpc_riccati.hpp).  tools/riccati_form_study.py measures what it does to the iterates (nothing at qp_tol = 1e-5).
 */
static void riccati_factor_classical(const orc_qp *qp, ipm_ws *w)
{
    const int N = qp->N;
    for (int i = 0; i < ORC_NX; i++) for (int j = 0; j < ORC_NX; j++) w->P[N][i][j] = w->Hh[N][ORC_NU + (i > j ? i : j)][ORC_NU + (i > j ? j : i)];
    for (int k = N - 1; k >= 0; k--) {
        double PBA[ORC_NX][ORC_NV], F[ORC_NV][ORC_NV];
        /* P_{k+1} enters unfactorised: a diagonal entry that cancellation has driven negative would go unnoticed -- the square-root form's pivot
         * tests would have caught it -- so the sign bits of the five diagonal entries are tested (the kernels test the same bits, csrc/tmpc_riccati.hpp) */
        for (int i = 0; i < ORC_NX; i++) if (signbit(w->P[k + 1][i][i])) w->bad = 1;
        for (int n = 0; n < ORC_NX; n++)
            for (int j = 0; j < ORC_NV; j++) {
                double acc = 0.0;
                for (int m = 0; m < ORC_NX; m++) acc += w->P[k + 1][n][m] * qp->BA[k][m][j];
                PBA[n][j] = acc;                                  /* (P [B A])_nj : column j is what the lane of row j computes */
            }
        for (int i = 0; i < ORC_NV; i++)
            for (int j = 0; j <= i; j++) {
                double acc = w->Hh[k][i][j];
                for (int n = 0; n < ORC_NX; n++) acc += PBA[n][i] * qp->BA[k][n][j];
                F[i][j] = acc;
            }
        memset(w->L[k], 0, sizeof w->L[k]);
        for (int c = 0; c < ORC_NU; c++) {                        /* right-looking elimination of the two input columns */
            double d = F[c][c];
            if (!(d > 0.0)) w->bad = 1;
            double y = 1.0 / sqrt(d);
            for (int i = c; i < ORC_NV; i++) w->L[k][i][c] = F[i][c] * y;
            for (int i = c + 1; i < ORC_NV; i++)
                for (int j = c + 1; j <= i; j++) F[i][j] -= w->L[k][i][c] * w->L[k][j][c];
        }
        for (int i = 0; i < ORC_NX; i++)
            for (int j = 0; j <= i; j++) w->P[k][i][j] = w->P[k][j][i] = F[ORC_NU + i][ORC_NU + j];
    }
}

static void riccati_factor(const orc_qp *qp, ipm_ws *w)
{
    const int N = qp->N;
    if (w->classical) { riccati_factor_classical(qp, w); return; }
    /* terminal: Lp = chol(Hh_N,xx) */
    {
        double A[ORC_NX][ORC_NX];
        for (int i = 0; i < ORC_NX; i++) for (int j = 0; j < ORC_NX; j++) A[i][j] = w->Hh[N][ORC_NU + i][ORC_NU + j];
        for (int j = 0; j < ORC_NX; j++) {
            double d = A[j][j];
            for (int l = 0; l < j; l++) d -= w->Lp[N][j][l] * w->Lp[N][j][l];
            if (!(d > 0.0)) w->bad = 1;
            double ljj = sqrt(d);
            w->Lp[N][j][j] = ljj;
            for (int i = j + 1; i < ORC_NX; i++) {
                double a = A[i][j];
                for (int l = 0; l < j; l++) a -= w->Lp[N][i][l] * w->Lp[N][j][l];
                w->Lp[N][i][j] = a / ljj;
            }
            for (int i = 0; i < j; i++) w->Lp[N][i][j] = 0.0;
        }
    }
    for (int k = N - 1; k >= 0; k--) {
        double G[ORC_NX][ORC_NV], F[ORC_NV][ORC_NV];
        for (int i = 0; i < ORC_NX; i++)
            for (int j = 0; j < ORC_NV; j++) {
                double acc = 0.0;
                for (int l = i; l < ORC_NX; l++) acc += w->Lp[k + 1][l][i] * qp->BA[k][l][j];   /* (Lp^T)_{il} = Lp_{li} */
                G[i][j] = acc;
            }
        for (int i = 0; i < ORC_NV; i++)
            for (int j = 0; j <= i; j++) {
                double acc = w->Hh[k][i][j];
                for (int l = 0; l < ORC_NX; l++) acc += G[l][i] * G[l][j];
                F[i][j] = acc;
            }
        for (int j = 0; j < ORC_NV; j++) {
            double d = F[j][j];
            for (int l = 0; l < j; l++) d -= w->L[k][j][l] * w->L[k][j][l];
            if (!(d > 0.0)) w->bad = 1;
            double ljj = sqrt(d);
            w->L[k][j][j] = ljj;
            for (int i = j + 1; i < ORC_NV; i++) {
                double a = F[i][j];
                for (int l = 0; l < j; l++) a -= w->L[k][i][l] * w->L[k][j][l];
                w->L[k][i][j] = a / ljj;
            }
            for (int i = 0; i < j; i++) w->L[k][i][j] = 0.0;
        }
        for (int i = 0; i < ORC_NX; i++)
            for (int j = 0; j < ORC_NX; j++) w->Lp[k][i][j] = w->L[k][ORC_NU + i][ORC_NU + j];
    }
}

/* y = P_{k} r = Lp (Lp^T r) */
static void apply_P(const double Lp[ORC_NX][ORC_NX], const double *r, double *y)
{
    double tmp[ORC_NX];
    for (int i = 0; i < ORC_NX; i++) { double a = 0.0; for (int l = i; l < ORC_NX; l++) a += Lp[l][i] * r[l]; tmp[i] = a; }
    for (int i = 0; i < ORC_NX; i++) { double a = 0.0; for (int l = 0; l <= i; l++) a += Lp[i][l] * tmp[l]; y[i] = a; }
}

static void apply_Pm(const double P[ORC_NX][ORC_NX], const double *r, double *y)
{
    for (int i = 0; i < ORC_NX; i++) { double a = 0.0; for (int l = 0; l < ORC_NX; l++) a += P[i][l] * r[l]; y[i] = a; }
}

static void riccati_solve(const orc_qp *qp, ipm_ws *w)
{
    const int N = qp->N;
    for (int i = 0; i < ORC_NX; i++) w->p[N][i] = w->gh[N][ORC_NU + i];
    for (int k = N - 1; k >= 0; k--) {
        double Pb[ORC_NX], f[ORC_NV];
        if (w->classical) apply_Pm(w->P[k + 1], w->rb[k], Pb); else
        apply_P(w->Lp[k + 1], w->rb[k], Pb);
        for (int i = 0; i < ORC_NX; i++) Pb[i] += w->p[k + 1][i];
        for (int j = 0; j < ORC_NV; j++) {
            double acc = w->gh[k][j];
            for (int l = 0; l < ORC_NX; l++) acc += qp->BA[k][l][j] * Pb[l];
            f[j] = acc;
        }
        /* y = Luu^-1 f_u ;  p_k = f_x - Lxu y */
        double y0 = f[0] / w->L[k][0][0];
        double y1 = (f[1] - w->L[k][1][0] * y0) / w->L[k][1][1];
        w->y[k][0] = y0; w->y[k][1] = y1;
        for (int i = 0; i < ORC_NX; i++)
            w->p[k][i] = f[ORC_NU + i] - w->L[k][ORC_NU + i][0] * y0 - w->L[k][ORC_NU + i][1] * y1;
    }
    double dx[ORC_NX] = {0};
    for (int k = 0; k < N; k++) {
        /* du = -Luu^-T (Lxu^T dx + y) */
        double r0 = w->y[k][0], r1 = w->y[k][1];
        for (int j = 0; j < ORC_NX; j++) { r0 += w->L[k][ORC_NU + j][0] * dx[j]; r1 += w->L[k][ORC_NU + j][1] * dx[j]; }
        double u1 = -r1 / w->L[k][1][1];
        double u0 = (-r0 - w->L[k][1][0] * u1) / w->L[k][0][0];
        w->dv[k][0] = u0; w->dv[k][1] = u1;
        for (int j = 0; j < ORC_NX; j++) w->dv[k][ORC_NU + j] = dx[j];
        double dxn[ORC_NX];
        for (int i = 0; i < ORC_NX; i++) {
            double acc = w->rb[k][i];
            for (int j = 0; j < ORC_NV; j++) acc += qp->BA[k][i][j] * w->dv[k][j];
            dxn[i] = acc;
        }
        if (w->classical) apply_Pm(w->P[k + 1], dxn, w->dpi[k + 1]); else
        apply_P(w->Lp[k + 1], dxn, w->dpi[k + 1]);
        for (int i = 0; i < ORC_NX; i++) w->dpi[k + 1][i] += w->p[k + 1][i];
        memcpy(dx, dxn, sizeof dx);
    }
    w->dv[N][0] = w->dv[N][1] = 0.0;
    for (int j = 0; j < ORC_NX; j++) w->dv[N][ORC_NU + j] = dx[j];
}

/* gh = rg + sum_i sgn c (q + lam rd)/t ; then Riccati vector solve; then dt, dlam */
static void newton_direction(const orc_qp *qp, const orc_qp_sol *s, ipm_ws *w)
{
    const int N = qp->N;
    for (int k = 0; k <= N; k++) {
        for (int j = 0; j < ORC_NV; j++) w->gh[k][j] = w->rg[k][j];
        for (int i = 0; i < qp->nrow[k]; i++) {
            double coef = qp->sgn[k][i] * (w->q[k][i] + s->lam[k][i] * w->rd[k][i]) / s->t[k][i];
            for (int j = 0; j < ORC_NV; j++) w->gh[k][j] += coef * qp->C[k][i][j];
        }
    }
    riccati_solve(qp, w);
    for (int k = 0; k <= N; k++)
        for (int i = 0; i < qp->nrow[k]; i++) {
            double cdv = 0.0;
            for (int j = 0; j < ORC_NV; j++) cdv += qp->C[k][i][j] * w->dv[k][j];
            w->dt[k][i] = qp->sgn[k][i] * cdv + w->rd[k][i];
            w->dlam[k][i] = -(w->q[k][i] + s->lam[k][i] * w->dt[k][i]) / s->t[k][i];
        }
}

static double max_step(const orc_qp *qp, const orc_qp_sol *s, const ipm_ws *w)
{
    double alpha = 1e300;
    for (int k = 0; k <= qp->N; k++)
        for (int i = 0; i < qp->nrow[k]; i++) {
            if (w->dt[k][i] < 0.0) { double a = -s->t[k][i] / w->dt[k][i]; if (a < alpha) alpha = a; }
            if (w->dlam[k][i] < 0.0) { double a = -s->lam[k][i] / w->dlam[k][i]; if (a < alpha) alpha = a; }
        }
    return alpha;
}

void orc_qp_solve(const orc_qp *qp, orc_qp_sol *s, int iter_max, double tol, double mu0, double thr0, double tau)
{
    orc_qp_solve_ex(qp, s, iter_max, tol, mu0, thr0, tau, 0, 0);
}

/* A box row has exactly one non-zero coefficient, 1.0 (build_qp writes them after the general rows). */
static int box_var(const orc_qp *qp, int k, int i)
{
    int var = -1;
    for (int j = 0; j < ORC_NV; j++) {
        if (qp->C[k][i][j] == 0.0) continue;
        if (qp->C[k][i][j] != 1.0 || var >= 0) return -1;
        var = j;
    }
    return var;
}

void orc_qp_solve_ex(const orc_qp *qp, orc_qp_sol *s, int iter_max, double tol, double mu0, double thr0, double tau, int warm, int init_box)
{
    orc_qp_solve_form(qp, s, iter_max, tol, mu0, thr0, tau, warm, init_box, 0);
}

void orc_qp_solve_form(const orc_qp *qp, orc_qp_sol *s, int iter_max, double tol, double mu0, double thr0, double tau, int warm, int init_box, int riccati_form)
{
    const int N = qp->N;
    /* per-thread workspace, reused across solves (a calloc/free pair per QP makes hundreds of OpenMP threads fight over
     * the kernel's page-fault path and says nothing about the algorithm) */
    static __thread ipm_ws *tls_w = NULL;
    if (!tls_w) tls_w = (ipm_ws *)malloc(sizeof(ipm_ws));
    ipm_ws *w = tls_w;
    memset(w, 0, sizeof(ipm_ws));
    w->classical = riccati_form == 1;
    int m = 0;
    /* cold start: v = 0 (dx_0 = given), pi = 0, t = max(residual, thr0), lam = mu0 / t.
     * warm >= 1: v of the previous QP (dx_0 is this QP's); warm == 2: pi, lam, t of the previous QP as well. */
    if (warm < 1) memset(s->v, 0, sizeof s->v);
    if (warm < 2) memset(s->pi, 0, sizeof s->pi);
    for (int j = 0; j < ORC_NX; j++) s->v[0][ORC_NU + j] = qp->dx0[j];
    s->v[N][0] = s->v[N][1] = 0.0;
    if (init_box && warm < 2) {
        /* HPIPM d_ocp_qp_init_var: a primal start that violates (or touches) a box by less than thr0 is moved inside by thr0, or to
         * the middle of the box when it is narrower than 2 thr0.  Box rows come in (lower, upper) pairs per variable. */
        for (int k = 0; k <= N; k++)
            for (int i = 0; i + 1 < qp->nrow[k]; i++) {
                const int var = box_var(qp, k, i);
                if (var < 0 || box_var(qp, k, i + 1) != var || qp->sgn[k][i] != 1.0 || qp->sgn[k][i + 1] != -1.0) continue;
                if (k == 0 && var >= ORC_NU) continue;                              /* dx_0 is fixed */
                const double lb = qp->beta[k][i], ub = qp->beta[k][i + 1];
                double x = s->v[k][var];
                if (x - lb < thr0) x = (ub - (lb + thr0) < thr0) ? 0.5 * (lb + ub) : lb + thr0;
                else if (ub - x < thr0) x = ub - thr0;
                s->v[k][var] = x;
                i++;
            }
    }
    for (int k = 0; k <= N; k++)
        for (int i = 0; i < qp->nrow[k]; i++, m++) {
            if (warm == 2) {
                /* a converged QP hands over t lam ~ 1e-10: started from there the method stalls on most QPs (measured: 14 % of the
                 * bench trajectories keep their full iteration budget with a 1e-12 floor, 95 % with 1e-6) -- the floor plays the role
                 * of HPIPM's t_min / lam_min arguments */
                if (!(s->t[k][i] > 1e-6)) s->t[k][i] = 1e-6;
                if (!(s->lam[k][i] > 1e-6)) s->lam[k][i] = 1e-6;
                continue;
            }
            double cv = 0.0;
            for (int j = 0; j < ORC_NV; j++) cv += qp->C[k][i][j] * s->v[k][j];
            double r = qp->sgn[k][i] * (cv - qp->beta[k][i]);
            s->t[k][i] = r > thr0 ? r : thr0;
            s->lam[k][i] = mu0 / s->t[k][i];
        }
    s->status = 2; s->iters = 0;
    for (int it = 0;; it++) {
        /* ---- residuals ---- */
        double res_g = 0.0, res_b = 0.0, res_d = 0.0, res_m = 0.0, mu = 0.0;
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < ORC_NV; i++) {
                double acc = qp->g[k][i];
                for (int j = 0; j < ORC_NV; j++) acc += qp->W[k][i][j] * s->v[k][j];
                if (k < N) for (int l = 0; l < ORC_NX; l++) acc += qp->BA[k][l][i] * s->pi[k + 1][l];
                if (i >= ORC_NU && k >= 1) acc -= s->pi[k][i - ORC_NU];
                w->rg[k][i] = acc;
            }
            for (int i = 0; i < qp->nrow[k]; i++) {
                double cv = 0.0;
                for (int j = 0; j < ORC_NV; j++) cv += qp->C[k][i][j] * s->v[k][j];
                for (int j = 0; j < ORC_NV; j++) w->rg[k][j] -= qp->sgn[k][i] * s->lam[k][i] * qp->C[k][i][j];
                w->rd[k][i] = qp->sgn[k][i] * (cv - qp->beta[k][i]) - s->t[k][i];
                double comp = s->lam[k][i] * s->t[k][i];
                mu += comp;
                if (fabs(w->rd[k][i]) > res_d) res_d = fabs(w->rd[k][i]);
                if (comp > res_m) res_m = comp;
            }
            if (k == N) w->rg[k][0] = w->rg[k][1] = 0.0;               /* no inputs at the terminal node */
            if (k == 0) for (int i = ORC_NU; i < ORC_NV; i++) w->rg[k][i] = 0.0;   /* dx_0 is fixed */
            for (int i = 0; i < ORC_NV; i++) if (fabs(w->rg[k][i]) > res_g) res_g = fabs(w->rg[k][i]);
            if (k < N)
                for (int i = 0; i < ORC_NX; i++) {
                    double acc = qp->b[k][i] - s->v[k + 1][ORC_NU + i];
                    for (int j = 0; j < ORC_NV; j++) acc += qp->BA[k][i][j] * s->v[k][j];
                    w->rb[k][i] = acc;
                    if (fabs(acc) > res_b) res_b = fabs(acc);
                }
        }
        mu = m > 0 ? mu / m : 0.0;
        if (getenv("ORC_IPM_TRACE")) fprintf(stderr, "ipm it %d res_g %.17g res_b %.17g res_d %.17g res_m %.17g mu %.17g\n", it, res_g, res_b, res_d, res_m, mu);
        if (!(isfinite(res_g) && isfinite(res_b) && isfinite(res_d) && isfinite(res_m))) { s->status = 4; break; }
        if (res_g <= tol && res_b <= tol && res_d <= tol && res_m <= tol) { s->status = 0; break; }
        if (it >= iter_max) { s->status = 2; break; }
        s->iters = it + 1;

        /* ---- barrier-augmented Hessian + factorisation ---- */
        for (int k = 0; k <= N; k++) {
            memcpy(w->Hh[k], qp->W[k], sizeof w->Hh[k]);
            for (int i = 0; i < qp->nrow[k]; i++) {
                double d = s->lam[k][i] / s->t[k][i];
                for (int a = 0; a < ORC_NV; a++)
                    for (int b = 0; b < ORC_NV; b++) w->Hh[k][a][b] += d * qp->C[k][i][a] * qp->C[k][i][b];
            }
        }
        w->bad = 0;
        riccati_factor(qp, w);
        if (w->bad) { if (getenv("ORC_IPM_TRACE")) fprintf(stderr, "riccati: Fuu not PD\n"); s->status = 4; break; }

        /* ---- predictor (sigma = 0) ---- */
        for (int k = 0; k <= N; k++) for (int i = 0; i < qp->nrow[k]; i++) w->q[k][i] = s->lam[k][i] * s->t[k][i];
        newton_direction(qp, s, w);
        double a_aff = max_step(qp, s, w); if (a_aff > 1.0) a_aff = 1.0;
        double mu_aff = 0.0;
        for (int k = 0; k <= N; k++)
            for (int i = 0; i < qp->nrow[k]; i++)
                mu_aff += (s->lam[k][i] + a_aff * w->dlam[k][i]) * (s->t[k][i] + a_aff * w->dt[k][i]);
        mu_aff = m > 0 ? mu_aff / m : 0.0;
        double sigma = mu > 0.0 ? (mu_aff / mu) : 0.0; sigma = sigma * sigma * sigma;

        /* ---- corrector ---- */
        for (int k = 0; k <= N; k++)
            for (int i = 0; i < qp->nrow[k]; i++)
                w->q[k][i] = s->lam[k][i] * s->t[k][i] - sigma * mu + w->dt[k][i] * w->dlam[k][i];
        newton_direction(qp, s, w);
        double alpha = tau * max_step(qp, s, w); if (alpha > 1.0) alpha = 1.0;
        if (getenv("ORC_IPM_TRACE")) fprintf(stderr, "   a_aff %.3e sigma %.3e alpha %.3e\n", a_aff, sigma, alpha);
        if (!isfinite(alpha)) { s->status = 4; break; }
        if (alpha < 1e-12) { s->status = 3; break; }

        for (int k = 0; k <= N; k++) {
            for (int j = 0; j < ORC_NV; j++) s->v[k][j] += alpha * w->dv[k][j];
            if (k >= 1) for (int j = 0; j < ORC_NX; j++) s->pi[k][j] += alpha * w->dpi[k][j];
            for (int i = 0; i < qp->nrow[k]; i++) {
                s->lam[k][i] += alpha * w->dlam[k][i];
                s->t[k][i] += alpha * w->dt[k][i];
            }
        }
    }
}
