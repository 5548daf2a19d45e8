/*
 * oracle/batch.c -- CPU ORACLE (test infrastructure).
 *
 * Batch driver restating the OpenMP loop of GuidanceConstraints::optimize
 * (mpc_planner_modules/src/guidance_constraints.cpp:279-361): one independent solve per local planner,
 * `#pragma omp parallel for`, then FindBestPlanner (:416-434).
 */
#include <stddef.h>
#include "tmpc_oracle.h"

void orc_solve_batch(const orc_problem *pb, int B, const double *xinit, const double *x0,
                     const double *params, double *xtraj, double *utraj, orc_info *info, int num_threads)
{
    const size_t n_x0 = (size_t)(pb->N + 1) * ORC_NVE, n_par = (size_t)pb->N * pb->npar;
    const size_t n_xt = (size_t)(pb->N + 1) * ORC_NXE, n_ut = (size_t)pb->N * ORC_NU;
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int b = 0; b < B; b++)
        orc_solve(pb, &xinit[(size_t)b * ORC_NXE], &x0[b * n_x0], &params[b * n_par],
                  &xtraj[b * n_xt], &utraj[b * n_ut], &info[b]);
}

int orc_find_best(int B, const double *objective, const int *exit_code, const unsigned char *disabled)
{
    double best_solution = 1e10;
    int best_index = -1;
    for (int i = 0; i < B; i++) {
        if (disabled && disabled[i]) continue;
        if (exit_code[i] == 1 && objective[i] < best_solution) { best_solution = objective[i]; best_index = i; }
    }
    return best_index;
}

/* Same batch loop, additionally recording the interior-point iterations of every QP: qp_iters[B][n_sqp] (0 = QP not reached). */
void orc_set_qp_iter_trace(int *per_sqp_iteration);
void orc_solve_batch_trace(const orc_problem *pb, int B, const double *xinit, const double *x0,
                           const double *params, double *xtraj, double *utraj, orc_info *info, int num_threads, int *qp_iters)
{
    const size_t n_x0 = (size_t)(pb->N + 1) * ORC_NVE, n_par = (size_t)pb->N * pb->npar;
    const size_t n_xt = (size_t)(pb->N + 1) * ORC_NXE, n_ut = (size_t)pb->N * ORC_NU;
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        for (int i = 0; i < pb->n_sqp; i++) qp_iters[(size_t)b * pb->n_sqp + i] = 0;
        orc_set_qp_iter_trace(&qp_iters[(size_t)b * pb->n_sqp]);
        orc_solve(pb, &xinit[(size_t)b * ORC_NXE], &x0[b * n_x0], &params[b * n_par],
                  &xtraj[b * n_xt], &utraj[b * n_ut], &info[b]);
        orc_set_qp_iter_trace(0);
    }
}
