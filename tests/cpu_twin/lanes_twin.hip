// tests/cpu_twin/lanes_twin.hip -- TEST INFRASTRUCTURE, not product code.
//
// Host instantiation of the lane-per-trajectory kernel's per-lane program (mpc_planner_amd/csrc/tmpc_lanes.hpp): the same
// scalar source the device kernel runs, executed lane by lane on the CPU over the same lane-major workspace layout.  Lets
// tests/ compare that program with the oracle here, without a GPU (algorithm + workspace layout; the device build differs only
// in its rsq/rcp seeds and FMA contraction).  Built by __graft_entry__.build() with `hipcc --cuda-host-only`; loaded by
// tests/test_lanes_twin.py only.  The product library never contains or calls this.
#include <stdlib.h>
#include <string.h>
#include "../../include/tmpc_hip.h"
#include "../../mpc_planner_amd/csrc/tmpc_lanes.hpp"

using namespace tmpc;
using namespace tmpc::lanes;

extern "C" int lanes_twin_solve(const tmpc_dims *dims, int32_t B, const double *xinit, const double *x0, const double *params,
                                double *xtraj, double *utraj, double *pobj, int32_t *exit_code, int32_t *qp_status,
                                int32_t *sqp_iter, double *res_eq, int32_t *qp_iter)
{
    Dims d{};
    d.N = dims->N; d.S = dims->S; d.n_lin = dims->n_lin; d.M = dims->M; d.npar = dims->npar;
    d.n_slk = dims->n_slk; d.slack = dims->slack;
    d.n_sqp = dims->n_sqp; d.qp_iter_max = dims->qp_iter_max; d.erk_steps = dims->erk_steps;
    d.dt = dims->dt; d.qp_tol = dims->qp_tol; d.reg_eps = dims->reg_eps; d.mu0 = dims->ipm_mu0; d.thr0 = dims->ipm_thr0;
    for (int i = 0; i < NV; i++) { d.lb[i] = dims->lb[i]; d.ub[i] = dims->ub[i]; }
    derive_dims(d);
    const Layout L = make_layout(d);
    const int nb = (B + LW - 1) / LW, N = d.N, nve = ext_nv(d), nxe = ext_nx(d);
    double *ws = (double *)calloc((size_t)nb * block_doubles(L), sizeof(double));
    if (!ws) return -1;
    for (int b = 0; b < B; b++) {                       // what lanes_transpose_in_kernel + lanes_prepare_kernel do
        double *w = ws + (size_t)(b / LW) * block_doubles(L) + (b % LW);
        for (int e = 0; e < N * d.npar; e++) w[((size_t)L.o_par + e) * LW] = params[(size_t)b * N * d.npar + e];
        for (int i = 0; i < nxe; i++) w[((size_t)L.o_xinit + i) * LW] = xinit[(size_t)b * nxe + i];
        for (int k = 0; k <= N; k++)
            for (int i = 0; i < NV; i++) w[((size_t)k * L.sd + L.o_z + i) * LW] = x0[((size_t)b * (N + 1) + k) * nve + i];
        w[((size_t)N * L.sd + L.o_z + 0) * LW] = 0.0; w[((size_t)N * L.sd + L.o_z + 1) * LW] = 0.0;      // (lanes_prepare_kernel)
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        const Lane ln{d, L, ws + (size_t)(b / LW) * block_doubles(L), (unsigned)(b % LW), nullptr};
        const Result R = ln.solve(d.n_sqp);
        const double sl = ln.slack();
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < NX; i++) xtraj[((size_t)b * (N + 1) + k) * nxe + i] = ln.F(k, L.o_z + NU + i);
            if (nxe > NX) xtraj[((size_t)b * (N + 1) + k) * nxe + NX] = sl;
        }
        for (int k = 0; k < N; k++)
            for (int i = 0; i < NU; i++) utraj[((size_t)b * N + k) * NU + i] = ln.F(k, L.o_z + i);
        pobj[b] = R.pobj; res_eq[b] = R.res_eq; exit_code[b] = R.exit_code;
        qp_status[b] = R.qp_status; sqp_iter[b] = R.sqp_iter; qp_iter[b] = R.qp_iter;
    }
    free(ws);
    return 0;
}
