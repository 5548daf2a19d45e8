// mpc_planner_amd/csrc/tmpc_lanes_api.hpp -- host interface between the C-ABI (tmpc_solve.hip) and the lane-per-trajectory
// throughput kernels (tmpc_lanes.hip).  Internal to the library; the public boundary stays include/tmpc_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "tmpc_stage.hpp"

namespace tmpc {
namespace lanes {

struct Context;

// Workspace for up to B_max trajectories on the current device (HBM: (N + 1) stage records + the transposed parameters).
Context *create(const Dims &d, int B_max, std::string &err);
void destroy(Context *c);
size_t workspace_bytes(const Context *c);

// Reference-layout inputs (device pointers: xinit [B][nx], x0 [B][(N+1) nvar], params [B][N npar]) -> lane-major workspace,
// transposed through LDS (coalesced reads and writes).  load_iterate: take the primal iterate from x0 (loadWarmstart); otherwise
// the iterate the workspace holds continues.  fresh_multipliers: zero them (a new solver instance); otherwise the multipliers
// of the previous call stay (the reference's capsules keep them across ticks).
int stage_in(Context *c, hipStream_t stream, int B, const double *xinit, const double *x0, const double *params, bool load_iterate,
             bool fresh_multipliers, std::string &err);
// n_iter RTI iterations for every trajectory + completeOneIteration; outputs in the reference layouts (device pointers).
// persistent: the tmpc_solve_iterations protocol (slots whose loop ended are skipped); complete: failed slots get zero multipliers.
int solve(Context *c, hipStream_t stream, int B, int n_iter, bool persistent, bool complete, double *xtraj, double *utraj, double *pobj, int *exit_code,
          int *qp_status, int *sqp_iter, double *res_eq, int *qp_iter, std::string &err);
int reset_multipliers(Context *c, hipStream_t stream, int B, std::string &err);
// A new solve() of the slots' Solvers: the "iteration loop has ended" marks of the previous solve are cleared (TMPC_ITER_NEW_SOLVE).
int clear_stopped(Context *c, hipStream_t stream, int B, std::string &err);

}  // namespace lanes
}  // namespace tmpc
