/*
 * include/tmpc_hip.h -- C-ABI of the MI355X batched SQP/NLP solve path (libtmpc_hip.so).
 *
 * This is the drop-in boundary under tud-amr/mpc_planner's `MPCPlanner::Solver`
 * (mpc_planner_solver/include/mpc_planner_solver/acados_solver_interface.h:93-222).  It replaces the inner
 * acados C ABI the reference's Solver calls (all call sites in
 * mpc_planner_solver/src/acados_solver_interface.cpp) with ONE batch-first interface: B independent
 * trajectories (= B reference `Solver` instances, i.e. the `planners_` loop of
 * mpc_planner_modules/src/guidance_constraints.cpp:279-361) are solved by one kernel launch.
 *
 * Plain C: opaque handle, plain pointers and sizes, int error codes.  No torch / HIP types in signatures
 * (device pointers travel as void*), no CUDA-compat headers.
 *
 * Layouts (identical to the reference's host structs, so `Solver::_params` can be passed as-is):
 *   xinit  [B][nx]            AcadosParameters::xinit            (acados_solver_interface.h:53)
 *   x0     [B][(N+1)*nvar]    AcadosParameters::x0, [u_k; x_k]   (:54)
 *   params [B][N*npar]        AcadosParameters::all_parameters   (:56), row k = stage k, node N reuses row N-1
 *   xtraj  [B][(N+1)*nx]      AcadosOutput::xtraj                (:129)
 *   utraj  [B][N*nu]          AcadosOutput::utraj                (:130)
 * nx = 5, nu = 2, nvar = 7 (ContouringSecondOrderUnicycleModel, solver_generator/solver_model.py:193-214), or
 * nx = 6, nvar = 8 with dims.slack = 1 (ContouringSecondOrderUnicycleModelWithSlack, solver_model.py:274-298: the
 * slack state is the last column).  acados pins that state (x_0 = xinit covers it and slack' = 0), so the solve
 * treats it as the constant xinit[5]: the slack column of x0 is ignored, xtraj returns xinit[5] in it.
 *
 * Error convention: every function returns 0 on success, <0 on error (TMPC_ERR_*); tmpc_last_error()
 * gives the message.  Per-trajectory solver outcomes use the reference's Forces-style exit codes
 * (acados_solver_interface.cpp:197-203, 391-424): 1 success, 0 generic failure, 2 max iter, 3 min step,
 * 4 QP failure; qp_status 0 ok / 2 max iter / 3 min step / 4 NaN.
 */
#ifndef TMPC_HIP_H
#define TMPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMPC_NU 2
#define TMPC_NX 5
#define TMPC_NV 7

#define TMPC_OK 0
#define TMPC_ERR_INVALID (-1)      /* bad argument / unsupported dimensions */
#define TMPC_ERR_HIP (-2)          /* HIP runtime error (message in tmpc_last_error) */
#define TMPC_ERR_NO_DEVICE (-3)    /* no gfx950 device / kernels missing */

/* Problem dimensions + solver options.  Replaces the compile-time SOLVER_* macros of the generated
 * acados solver and the options of solver_generator/generate_acados_solver.py:143-177. */
typedef struct tmpc_dims {
    int32_t N;            /* horizon (settings.yaml "N"); nodes 0..N */
    int32_t S;            /* contouring/num_segments */
    int32_t n_lin;        /* topology halfspace rows per stage (max_obstacles + add_halfspaces; 0 = no guidance module) */
    int32_t M;            /* ellipsoid rows per stage (max_obstacles, n_discs = 1) */
    int32_t npar;         /* parameters per stage; must equal the reference's count for these modules:
                             8 + slack + 9 S + 3 n_lin + (M ? 2 + 7 M : 0) + (n_slk ? (M ? 0 : 1) + 3 n_slk : 0) */
    int32_t n_sqp;        /* solver_settings/acados/iterations (RTI iterations per solve) */
    int32_t qp_iter_max;  /* qp_solver_iter_max = 50 */
    int32_t erk_steps;    /* sim_method_num_steps = 3 (ERK4) */
    double dt;            /* integrator_step */
    double qp_tol;        /* qp_tol = 1e-5 */
    double reg_eps;       /* MIRROR epsilon (acados default 1e-4) */
    double ipm_mu0;       /* interior-point initial barrier (0.01) */
    double ipm_thr0;      /* interior-point initial slack floor (0.01) */
    double lb[TMPC_NV];   /* model bounds, order [a,w,x,y,psi,v,spline] (solver_model.py:204-205) */
    double ub[TMPC_NV];
    int32_t n_slk;        /* decomp / scenario halfspace rows a1 x + a2 y - (b + slack) <= 0 per stage
                             (decomp_constraints.py:68-98, scenario_constraints.py:64-94); scenario rows first */
    int32_t slack;        /* 1: slack model (nx = 6, nvar = 8; MPCBase weighs the slack state) */
    int32_t cost_model;   /* 0: ContouringModule (MPCC: contouring.py:48-98); 1: CurvatureAwareContouringModule (CA-MPC:
                             curvature_aware_contouring.py:48-105; same parameter map, spline ODE s' = v -- BASELINE configs[2]).  Set by
                             tmpc_default_dims* to 0.  Hand-written kernels only (a generated solver's cost is its module stack's). */
    int32_t row_model;    /* what the M obstacle rows are.  0: EllipsoidConstraintModule (ellipsoid_constraints.py:66-110: 7 parameters per
                             obstacle, h >= 1); 1: GaussianConstraintModule (gaussian_constraints.py:33-113: 6 parameters per obstacle -- x, y,
                             major, minor, risk, r --, h >= 0; the collision-avoidance submodule of mpc_planner_jackal's default T-MPC,
                             generate_jackal_solver.py:53-73).  npar follows: 6 instead of 7 entries per obstacle.  Set by tmpc_default_dims* to 0.
                             Together with cost_model 1 (round 6): the generic kernel, and the four-wave tick kernel for 21 <= N <= 31.
                             Hand-written kernels only. */
    int32_t riccati_form; /* form of the Riccati recursion inside the interior-point QP solver (round 6).  TMPC_RICCATI_SCHUR (0, what
                             tmpc_default_dims* sets): the cost-to-go Hessian P_k is kept as the Schur complement F_xx - Lxu Lxu^T (HPIPM's
                             square_root_alg = 0 [UPSTREAM]) -- every kernel family, every latency mode.  TMPC_RICCATI_SQUARE_ROOT (1): P_k is
                             re-factorised at every stage (P_k = Lxx Lxx^T: HPIPM's square_root_alg = 1 [UPSTREAM], the default of the mode
                             acados configures, solver_generator/generate_acados_solver.py:171) -- the run-time-shape fast kernels only (N <= 32,
                             any row mix of cost_model 0; cost_model 1 at N > 20; no compact / latency variants: tmpc_set_latency_mode answers 1).
                             SUPPORTED qp_tol RANGE: at qp_tol >= 1e-7 the two forms give the same exit codes, iteration counts and iterates to
                             rounding (profiles/round5_riccati_form_study.json: 0 of 6400 solves differ at the reference's 1e-5); below that the
                             interior-point method runs into the conditioning of the barrier systems and 0.4-3.4 % of the solves end differently
                             at 1e-9 -- a comparison with a real acados at such a tolerance (tools/acados_replay.py --qp-tol 1e-9) should run form 1.
                             (The CPU oracle numbers its option the other way round: orc_problem.riccati_form 0 = square-root, 1 = Schur.) */
} tmpc_dims;
#define TMPC_RICCATI_SCHUR 0
#define TMPC_RICCATI_SQUARE_ROOT 1

typedef struct tmpc_handle tmpc_handle;

/* Defaults for the Jackal contouring unicycle (settings.yaml + generate_acados_solver.py). */
void tmpc_default_dims(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M);
/* Same with decomp / scenario rows and the slack model (configuration_safe_horizon,
 * generate_jackalsimulator_solver.py:67-90; rosnavigation configuration_tmpc, generate_rosnavigation_solver.py:86-108). */
void tmpc_default_dims_ex(tmpc_dims *d, int32_t N, int32_t S, int32_t n_lin, int32_t M, int32_t n_slk, int32_t slack);

/* Replaces Solver_acados_create_capsule + Solver_acados_create_with_discretization
 * (acados_solver_interface.cpp:17,33) for B_max solver instances at once.  Owns device buffers + stream. */
int tmpc_create(tmpc_handle **out, const tmpc_dims *dims, int32_t B_max, int32_t device);
/* Same, for callers that may have been built against another revision of this header: `dims_size` = the caller's sizeof(tmpc_dims).  Fields
 * the caller's struct does not have (it is shorter: cost_model / row_model were appended in round 4, riccati_form in round 6) take their
 * defaults (0) instead of being read from whatever follows the caller's struct.  Accepted sizes are exactly the struct's REVISION BOUNDARIES
 * (the size up to n_slk / slack, up to row_model, the current one): a size that ends inside a field is refused.  A LONGER struct (a newer header
 * than this library) is accepted only if every byte beyond this library's struct is zero -- a non-zero field the library does not know is an
 * option it cannot honour, and is refused (TMPC_ERR_INVALID) instead of being ignored.  New code should call this one:
 * tmpc_create(out, dims, ..) == tmpc_create_v2(out, dims, sizeof(tmpc_dims), ..) of the SAME header revision. */
int tmpc_create_v2(tmpc_handle **out, const tmpc_dims *dims, uint32_t dims_size, int32_t B_max, int32_t device);
/* Replaces Solver_acados_free + Solver_acados_free_capsule (:54,60). */
void tmpc_destroy(tmpc_handle *h);
const char *tmpc_last_error(const tmpc_handle *h);

/* Replaces, for B solvers: ocp_nlp_constraints_model_set(lbx/ubx = xinit) (:124-125),
 * Solver_acados_update_params (:127-135) and loadWarmstart's ocp_nlp_out_set (:274-284).
 * Host pointers, copied H2D on the handle's stream. */
int tmpc_set_batch(tmpc_handle *h, int32_t B, const double *xinit, const double *x0, const double *params);
/* Same, inputs already resident in HBM (device pointers, same layouts); no copy is made. */
int tmpc_set_batch_device(tmpc_handle *h, int32_t B, const void *d_xinit, const void *d_x0, const void *d_params);

/* Replaces Solver::solve() = initializeOneIteration + n_sqp x solveOneIteration + completeOneIteration
 * (:86-204) for all B trajectories: one kernel launch, asynchronous on the handle's stream. */
int tmpc_solve(tmpc_handle *h);
int tmpc_synchronize(tmpc_handle *h);

/* ---- persistent solver state: the "one iteration at a time" protocol and multipliers carried across ticks ---------------
 * The reference's capsules keep the NLP iterate and its multipliers between calls: Solver::solveOneIteration() = ONE
 * Solver_acados_solve continuing from them (acados_solver_interface.cpp:149-160; driven by SH-MPC's scenario module,
 * scenario_constraints.cpp:85), Solver::operator= copies parameters only (:67-77), loadWarmstart() overwrites the primal iterate
 * only (:274-284), and a solve that does not succeed resets the capsule (:187-191).  tmpc_solve_iterations does n_iter RTI
 * iterations for every trajectory slot of the current batch and then completeOneIteration (so tmpc_get is valid after every
 * call), starting
 *   from the batch's warm start x0 and zero multipliers                      flags = 0            (a fresh capsule)
 *   from the iterate the handle holds for the slot (no loadWarmstart)         TMPC_ITER_KEEP_ITERATE
 *   with the multipliers the handle holds for the slot                        TMPC_ITER_KEEP_MULTIPLIERS
 *   and, as the last call of a solve (completeOneIteration, :162-204)          TMPC_ITER_COMPLETE: slots whose exit code is
 *                                                                             not 1 get zero multipliers (the capsule reset, :187-191)
 * and stores iterate + multipliers of every slot afterwards.  A slot whose QP stopped with qp_status != 0 has left the reference's iteration loop (:105-106): further
 * KEEP_ITERATE calls leave it untouched until a call without KEEP_ITERATE loads a new warm start or a call with TMPC_ITER_NEW_SOLVE starts the next solve.  Slots the handle has no state for yet (a batch larger than any before) start fresh whatever the flags say.  n_iter calls with one
 * iteration each give bitwise the same result as one call with n_iter.  The first call on a handle has nothing to keep and
 * behaves like flags = 0, and so does the first call after tmpc_set_throughput_mode changed the kernel family (the wave kernels keep
 * the state in per-slot arrays, the lane kernels in their workspace).  tmpc_solve() itself never reads or writes this state. */
#define TMPC_ITER_KEEP_ITERATE 1
#define TMPC_ITER_KEEP_MULTIPLIERS 2
#define TMPC_ITER_COMPLETE 4
/* first call of a NEW solve() of the slots' Solvers: a slot's "left the iteration loop" mark (a QP that stopped with qp_status != 0) belongs
 * to the solve that set it -- the reference's loop exit is local to one solve() (:105-106) and every new solve() iterates again from the
 * capsule's state, warm start loaded or not. */
#define TMPC_ITER_NEW_SOLVE 8
int tmpc_solve_iterations(tmpc_handle *h, int32_t n_iter, int32_t flags);
/* State slots.  By default batch entry b uses state slot b.  A caller that owns one slot per Solver (the reference: one capsule
 * per Solver, acados_solver_interface.cpp:17,51-65) but launches a changing subset of them -- GuidanceConstraints::optimize skips
 * disabled planners (guidance_constraints.cpp:286-293) -- passes the slot of every entry of the CURRENT batch: slots[B], distinct,
 * in [0, B_max).  The map stays in force for the following tmpc_solve_iterations calls until it is replaced or cleared
 * (slots = NULL).  A slot nothing was stored in yet starts like a fresh capsule whatever the keep-flags say.  Wave kernels only. */
int tmpc_set_slots(tmpc_handle *h, const int32_t *slots);
/* Parameter-sharing hint for the current batch (optional; results are bitwise the same with and without it).  base_of[b] (host, one
 * entry per batch entry, values in [0, B)): entry b's parameter rows equal entry base_of[b]'s except for its own topology and
 * scenario halfspace rows (LinearizedConstraints / ScenarioConstraints::setParameters) -- which is how a guidance set comes about:
 * every planner's solver starts as a copy of the main solver (guidance_constraints.cpp:300 `*solver = *_solver`) and the shared
 * modules write the same values into each (:312-318).  The kernels then read everything but those rows from entry base_of[b]: a set's
 * 64 copies of the obstacle / spline / weight rows are fetched once instead of 64 times (the parameter rows are 2/3 of the path's
 * algorithmic bytes, and at eight trajectories per CU they no longer fit the L2 next to the solve's workspace).  The CALLER guarantees
 * the equality -- or, equivalently for the kernels that honour the hint, simply does not maintain the copies: a caller that keeps its
 * parameter tensor on the device writes a tick's shared rows ONCE per set, into entry base_of[b] (what bench.py's end-to-end step does), and the
 * other entries' shared columns are never read.  (The lane kernels and generated solvers ignore the hint and read every entry's own rows: such
 * a caller must not use them.)  The map belongs to the batch it was given for: it stays in force until it is replaced, cleared (NULL) or a tmpc_set_batch* call
 * names new inputs (kernels of this library that rewrite a batch's rows in place -- tmpc_linearize_topology, tmpc_scenario_halfspaces --
 * touch the entries' own rows only and keep it valid).  Ignored by the lane kernels
 * (tmpc_set_throughput_mode) and by generated solvers. */
int tmpc_set_param_sharing(tmpc_handle *h, const int32_t *base_of);
/* The "copies are not maintained" mode, made explicit (round-4 advisor): with TMPC_SHARE_COPIES_NOT_MAINTAINED the caller declares that the
 * shared columns of entries b != base_of[b] hold NO valid data.  From then on every path that would read them is an error instead of a silent
 * fallback: tmpc_solve / tmpc_solve_iterations return TMPC_ERR_INVALID while the map is not in force for the current batch (every tmpc_set_batch*
 * drops it: give it again), tmpc_set_throughput_mode (lane kernels) and tmpc_debug_profile (profiled twins) refuse, and generated solvers refuse
 * the flag itself.  Cleared by a NULL map.  flags = 0: tmpc_set_param_sharing (a pure hint, copies equal). */
#define TMPC_SHARE_COPIES_NOT_MAINTAINED 1
int tmpc_set_param_sharing_ex(tmpc_handle *h, const int32_t *base_of, int32_t flags);
/* Copy the persistent state of min(B_max) slots from another handle of the same shape and device (a caller that outgrew its handle). */
int tmpc_copy_state(tmpc_handle *dst, tmpc_handle *src);
/* Forget the persistent state of ONE slot: its next tmpc_solve_iterations starts like a fresh capsule whatever the keep-flags say
 * (a caller that hands the slot of a destroyed Solver to a new one: a new acados capsule, acados_solver_interface.cpp:17-33). */
int tmpc_clear_slot(tmpc_handle *h, int32_t slot);
/* Zero the multipliers of every slot (a new capsule / Solver_acados_reset). */
int tmpc_reset_multipliers(tmpc_handle *h);
/* Kernel variant for the following tmpc_solve calls: 0 (default) = throughput variant, 1 = latency variant (two waves per
 * trajectory; for control ticks of a few planners, like the 8 OpenMP threads of guidance_constraints.cpp:279), 2 = latency
 * variant with the interior-point Newton systems solved parallel in time (multiplier Schur complement + block cyclic reduction
 * over the stages, csrc/tmpc_scan.hpp) instead of by the stage-by-stage Riccati recursion that acados / HPIPM -- and modes 0, 1 --
 * use: 30-40 % less kernel time per tick; available for horizons N <= 31 of the hand-written model, at one workgroup per CU (a launch of
 * more than 256 trajectories per GPU takes several rounds: the default kernels are the better choice there).  Returns 0, or 1 if the handle's shape has no such variant (mode 2 then runs as mode 1,
 * mode 1 as the default kernel).  A trajectory's result is bitwise independent of the batch it is solved in.  Modes 0 and 1 agree
 * to rounding (1e-11), not bitwise.  Mode 2 is another factorisation of the same systems: its steps agree with the recursion's to
 * ~1e-6 on ill-conditioned late iterations (each is that far from an exact solve), trajectories agree within the 1e-4 parity
 * tolerance, and the interior-point iteration count of a solve can differ by one where a residual sits at the tolerance.
 * 3 (round 6) = FOUR waves per trajectory, for the launches that leave a whole CU to each trajectory (<= one workgroup per CU: the
 * reference's deployed 4 + 1 planners, a 64-trajectory tick): the algorithm of mode 2 with the stage evaluation split four ways by content,
 * the row passes at twelve (N <= 20) or eight (21 <= N <= 31: the horizon the reference ships, N = 30) lanes per stage and the wide phases of
 * the factorisation on all 256 lanes -- 10-15 % less kernel time per tick than mode 2 (measured ticks: DESIGN.md section 5), results equal to
 * mode 2's to rounding; N <= 20: MPCC or Gaussian rows; 21 <= N <= 31: every stage model, up to 34 rows per stage.  A shape without it runs
 * mode 2 (return value 1).  tmpc_latency_mode_capacity says how many trajectories a variant serves well in one launch. */
int tmpc_set_latency_mode(tmpc_handle *h, int32_t on);
/* How many trajectories ONE launch of kernel variant `mode` (0 .. 3 as above) holds resident on this device (workgroups per CU x CUs; variant 3: ONE workgroup per CU, what it is built for), 0 if the
 * handle's shape has no such variant, < 0 on error.  A latency variant pays off while the launch fits (one dependent chain deep); above it the
 * throughput kernels win.  The library never switches variants by batch size (results would depend on the rest of the batch): the CALLER decides,
 * e.g. the C++ BatchContext (cpp/src/solver_interface.cpp) asks for the tick variant only for batches within this capacity. */
int tmpc_latency_mode_capacity(tmpc_handle *h, int32_t mode);
/* Throughput variant for large batches (many control ticks / scenario solvers per launch): one LANE per trajectory instead of
 * one wavefront -- every lane runs the scalar SQP_RTI program on its own trajectory, the per-trajectory state is streamed from a
 * lane-major HBM workspace (allocated for B_max trajectories on the first call: about 8 (N+1) (175 + 6 nh) + 8 N npar bytes
 * each), and the reference-layout inputs are transposed into it at the start of every tmpc_solve.  Same algorithm and stopping
 * rules as the default kernels: exit codes and iteration counts agree, trajectories agree to rounding (1e-10), and a
 * trajectory's result does not depend on the rest of the batch.  The default (0) stays the choice for control ticks of a few
 * planners; the mode is chosen by the caller, never by the batch size.  Returns 0, or <0 if the workspace cannot be allocated or if
 * this build of the lane kernel spills registers to scratch (possible in generated solvers: refused, tmpc_last_error says so). */
int tmpc_set_throughput_mode(tmpc_handle *h, int32_t on);
/* 1 if this build of the library contains the lane-per-trajectory kernels, 0 if not (the default build since round 5: the family is optional,
 * -DTMPC_WITH_LANES; tmpc_set_throughput_mode(h, 1) then fails with a message that says so). */
int tmpc_has_lane_kernels(void);

/* Replaces ocp_nlp_out_get / ocp_nlp_get / ocp_nlp_eval_cost of completeOneIteration (:162-204).
 * Any pointer may be NULL.  Synchronises the stream.  Host pointers. */
int tmpc_get(tmpc_handle *h, double *xtraj, double *utraj, double *pobj, int32_t *exit_code,
             int32_t *qp_status, int32_t *sqp_iter, double *res_eq, int32_t *qp_iter_total);

/* Replaces FindBestPlanner (guidance_constraints.cpp:416-434) on device for ONE planner set = the contiguous trajectories
 * [first, first+count) of the batch (one scene's local planners): argmin of pobj*weight among exit_code == 1 && !disabled; init 1e10,
 * strict '<' (lowest index wins ties).  Index convention: everything is RELATIVE TO `first`, like an index into the reference's
 * planners_ vector -- weight[i] / disabled[i] (host pointers, `count` entries, or NULL) belong to trajectory first + i, and *best
 * is in [0, count) (the batch index is first + *best), or -1 if none. */
int tmpc_select_best(tmpc_handle *h, int32_t first, int32_t count, const double *weight,
                     const uint8_t *disabled, int32_t *best);

/* Device-resident result records for the multi-GPU all-gather: pointers to the handle's
 * pobj[B] (f64) and exit_code[B] (i32) device arrays. */
int tmpc_result_device_ptrs(tmpc_handle *h, void **d_pobj, void **d_exit_code);
/* The HIP stream (hipStream_t) every launch of this handle is enqueued on, for callers that order their own work against it
 * without host synchronisation -- e.g. the multi-GPU step runs the record all-gather stream-ordered between tmpc_pack_records and
 * tmpc_select_best_records by making that stream the collective's current stream. */
int tmpc_get_stream(tmpc_handle *h, void **stream);
/* Which solve kernel this handle dispatches and how it is launched, as a short text for logs and benchmark records (kernel family,
 * trajectories per workgroup, LDS bytes per workgroup, resident workgroups of a persistent launch).  Returns the length written. */
int tmpc_kernel_info(const tmpc_handle *h, char *buf, int32_t capacity);

/* ---- multi-GPU sharding: a scene's trajectories are split over ranks; after the solve every rank
 * packs one 16-byte record per local trajectory, the host all-gathers the record arrays (RCCL over xGMI via
 * torch.distributed) and every rank runs the same deterministic selection over the gathered array. -------- */
typedef struct tmpc_record {
    double objective;     /* _info.pobj (x consistency weight if supplied) */
    int32_t exit_code;    /* SolverResult::exit_code (guidance_constraints.h:32-50) */
    int32_t guidance_id;  /* SolverResult::guidance_ID */
} tmpc_record;

/* Pack records of the handle's B trajectories into d_records[B] (device pointer, caller-owned).
 * d_guidance_id: device int32[B] or NULL (then the local trajectory index); d_weight: device f64[B] or NULL. */
int tmpc_pack_records(tmpc_handle *h, void *d_records, const void *d_guidance_id, const void *d_weight);

/* FindBestPlanner over gathered records laid out [n_ranks][n_scenes][per_rank] (device pointer): for each scene
 * the winner is the lowest GLOBAL index (rank*per_rank + t) among exit_code == 1 with the smallest objective
 * (init 1e10, strict '<').  d_best: device int32[n_scenes], -1 if none.  Runs on the handle's stream. */
int tmpc_select_best_records(tmpc_handle *h, const void *d_records, int32_t n_ranks, int32_t n_scenes,
                             int32_t per_rank, void *d_best);

/* The winners' trajectories in one compact buffer -- what GuidanceConstraints::optimize copies from the best planner into the main solver
 * (`_solver->_output = best_solver->_output`, guidance_constraints.cpp:382-384), for every set of the launch at once, so that ONE small
 * device-to-host copy brings a tick's (or many ticks') results back.  d_best: i32 [n_sets] as written by tmpc_select_best_records (index
 * inside the set, -1: no successful trajectory).  The batch holds the entries [index_offset, index_offset + set_size) of every set (one
 * rank's share; index_offset = rank * per_rank, 0 on one GPU): a winner outside that range is another rank's and its rows are left untouched,
 * a set without a winner gets NaNs.  d_xtraj f64 [n_sets][(N + 1) nx], d_utraj f64 [n_sets][N nu] (device).  Runs on the handle's stream. */
int tmpc_gather_best(tmpc_handle *h, const void *d_best, int32_t n_sets, int32_t set_size, int32_t index_offset, void *d_xtraj, void *d_utraj);

/* Per-launch timing: when enabled, every tmpc_solve is bracketed by HIP events recorded on the handle's stream.
 * tmpc_get_timings synchronises and returns the durations [ms] of the launches since enable/last read. */
int tmpc_enable_timing(tmpc_handle *h, int32_t max_records);
int tmpc_get_timings(tmpc_handle *h, float *ms, int32_t capacity, int32_t *n_out);

/* Timing helper: runs tmpc_solve `reps` times back-to-back on the handle's stream, each bracketed by HIP
 * events recorded on THAT stream, and returns the per-launch kernel durations in milliseconds. */
int tmpc_time_solve(tmpc_handle *h, int32_t reps, float *ms_each);

/* ---- next row f-1: LinearizedConstraints::update + setParameters on device -----------------------------------
 * For every trajectory b of the current batch, stage k = 1..N-1 and obstacle j < n_lin: project the guess position
 * x0[b][k].(x,y) out of the discs of radius (1e-3 + robot_radius) around the obstacle predictions (<= 3 sweeps,
 * LinearizedConstraints::projectToSafety, linearized_constraints.cpp:130-148 -- the Douglas-Rachford operator of
 * ros_tools is not in the reference tree; restated from the published operator, DESIGN.md U10 -- the identity for collision-free guesses), then
 * a = (o - p)/|o - p|, b = a.o - (1e-3 + robot_radius) (:84-105, guidance mode) and write lin_constraint_j_{a1,a2,b}
 * into params[b][k]; stage 0 and the rows of non-guided planners get the dummies (1, 0, state_x + 100) (:155-166,
 * guidance_constraints.cpp:301-305).  The batch's params buffer is modified IN PLACE (device memory).
 * ALL n_lin topology rows are treated as dynamic obstacles (d_obstacle_pos has n_lin entries per scene); tmpc_linearize_topology_ex
 * below covers fewer obstacles than rows, static halfspace rows and the disc mode.
 *   d_obstacle_pos : f64 [n_scenes][n_lin][N][2]   prediction step i of obstacle j (stage k uses step k-1)
 *   d_scene_of     : i32 [B]                       scene of trajectory b
 *   d_state_x      : f64 [n_scenes]                current state x (for the dummy b)
 *   d_is_original  : u8  [B] or NULL               1 = non-guided T-MPC++ planner (all rows dummy) */
int tmpc_linearize_topology(tmpc_handle *h, const void *d_obstacle_pos, const void *d_scene_of, const void *d_state_x,
                            double robot_radius, const void *d_is_original);
/* The whole of LinearizedConstraints::update / setParameters (linearized_constraints.cpp:49-189):
 *   rows 0 .. n_obstacles-1          dynamic obstacles, d_obstacle_pos f64 [n_scenes][n_obstacles][N][2]
 *   rows n_obstacles .. +n_static-1  static halfspaces of the stage, copied as given (`linearized_constraints/add_halfspaces`, :107-123):
 *                                    d_static_halfspaces f64 [n_scenes][N][n_static][3] = (a1, a2, b) for stage k (k = 0 unused)
 *   remaining rows up to n_lin       dummies (1, 0, state_x + 100) (:181-187); n_obstacles + n_static <= n_lin
 * d_obstacle_radius == NULL: guidance mode, every obstacle disc has radius 1e-3 + robot_radius (:99, :140).  Otherwise the
 * `_use_guidance == false` branch (:63-73): f64 [n_scenes][n_obstacles], obstacle j's own radius + robot_radius in the projection and
 * in b.  One disc at the robot's centre (n_discs = 1, offset 0: the Jackal configurations); rows of further discs are not generated. */
int tmpc_linearize_topology_ex(tmpc_handle *h, const void *d_obstacle_pos, int32_t n_obstacles, const void *d_obstacle_radius,
                               const void *d_static_halfspaces, int32_t n_static, const void *d_scene_of, const void *d_state_x,
                               double robot_radius, const void *d_is_original);

/* ---- scenario -> polygon construction of SH-MPC on device.  Replaces what the reference gets from the
 * external scenario_module (scenario_constraints.cpp:47 update, :76-79 setParameters; source absent -> restated, see
 * mpc_planner_amd/modules.py::scenario_halfspaces): for every trajectory b and stage k >= 1, each of the n_pts sampled
 * obstacle positions of prediction step k-1 gives a halfspace a.x <= b linearised around the guess x0[b][k]; the
 * halfspaces that form the boundary of their intersection polygon (all others are redundant) are written, closest first
 * and at most n_rows of them (<= 64), into the first n_rows decomp/scenario rows of the batch's parameter tensor together
 * with ego_disc_0_offset; unused rows and stage 0 = dummy rows.  n_pts <= about 5480 (a stage's halfspaces live in LDS next to the kernel's static tables; the call checks the exact bound).
 * Device pointers:
 *   d_samples  : f64 [n_scenes][N][n_pts][2]   sampled positions, n_pts = obstacles x scenarios (index i = step k-1)
 *   d_scene_of : i32 [B];  d_state_x : f64 [n_scenes]  (dummy b = x + 100)
 * Operates in place on the parameter tensor of the last tmpc_set_batch / tmpc_set_batch_device call. */
int tmpc_scenario_halfspaces(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_rows, const void *d_scene_of,
                             const void *d_state_x, double radius, double disc_offset);

/* Support of each trajectory's solution after a solve on rows written by tmpc_scenario_halfspaces: the number of distinct
 * scenarios with an active constraint, a.p_disc - (b + slack) >= -tol at the solution (ScenarioSolver::support,
 * scenario_constraints.h:38-40; filled by the absent scenario_module, restated from the method's definition: the plan's
 * collision-probability certificate holds while the support stays within the bound the sample size was chosen for -- see
 * mpc_planner_amd/modules.py::scenario_risk / scenario_sample_size).  Scenario of sample i = i % n_scenarios (samples are
 * [obstacle][scenario]); n_scenarios <= 8192.
 * Valid only while the batch's scenario rows are the ones tmpc_scenario_halfspaces wrote (a later tmpc_set_batch* invalidates
 * the bookkeeping: TMPC_ERR_INVALID).
 *   d_support     : i32 [B] out    distinct active scenarios
 *   d_active_rows : i32 [B] out or NULL    active rows */
int tmpc_scenario_support(tmpc_handle *h, int32_t n_scenarios, double tol, void *d_support, void *d_active_rows);
/* SH-MPC scenario sampler on device (f-3; scenario_constraints.cpp:121-131: scenario_module.GetSampler().IntegrateAndTranslateToMeanAndVariance
 * per solver -- the scenario_module is absent, the sampler is restated from the call's inputs): for each of n_solvers solvers (scenes), each of
 * n_obstacles obstacles and each of n_scenarios scenarios, a mode of the obstacle's Gaussian mixture is drawn from d_prob
 * [n_solvers][n_obstacles][n_modes] and ONE standard-normal pair places the obstacle on every prediction step of that mode:
 * o_k = mean_k + R(angle_k) (major_k xi1, minor_k xi2), d_pred [n_solvers][n_obstacles][n_modes][N][6] = (x, y, cos angle, sin angle, major,
 * minor).  d_samples [n_solvers][N][n_obstacles * n_scenarios][2] is what tmpc_scenario_halfspaces / tmpc_scenario_discard read.
 * Counter-based (splitmix64 of seed, solver, obstacle, scenario) and free of library transcendentals: mpc_planner_amd.modules.sample_scenarios
 * reproduces every sample bit for bit. */
int tmpc_sample_scenarios(tmpc_handle *h, const void *d_pred, const void *d_prob, int32_t n_solvers, int32_t n_obstacles, int32_t n_modes,
                          int32_t n_scenarios, uint64_t seed, void *d_samples);
/* Scenario removal: for every trajectory of the current batch the n_discard scenarios that constrain its guess most (smallest clearance
 * min over obstacles and stages of |o - p_k| - radius; lowest scenario index on ties) are marked; the next tmpc_scenario_halfspaces on this
 * batch leaves their samples out.  The discarded scenarios count into the bound (modules.scenario_risk(removed = n_discard)).  The policy of
 * the absent scenario_module is not in the reference tree: this is the greedy rule of the method the reference cites (README.md:22).
 * tmpc_scenario_discarded copies the marks (uint8 [B][n_scenarios], device). */
int tmpc_scenario_discard(tmpc_handle *h, const void *d_samples, int32_t n_pts, int32_t n_scenarios, int32_t n_discard, const void *d_scene_of, double radius);
int tmpc_scenario_discarded(tmpc_handle *h, void *d_mask);
/* Stages of every trajectory whose sampled halfspaces CONTRADICT each other (an empty polygon: the guess sits in the overlap of inflated
 * discs on opposite sides).  Such a stage keeps the n_rows closest halfspaces instead of dummies -- the QP is then infeasible or pays
 * slack, never silently unconstrained -- and is counted here: d_count int32 [B] (device).  Callers treat a trajectory with a count
 * > 0 as not eligible (solver.optimize_scenarios: scenario_status 2). */
int tmpc_scenario_empty_stages(tmpc_handle *h, void *d_count);

/* ---- cross-tick state on device, so a closed loop runs without host round trips --------------------
 * tmpc_warmstart builds the next tick's warm start x0 and xinit of every trajectory of the current batch from the
 * solution the handle holds (last tmpc_solve) and the new state.  Device pointers:
 *   d_state : f64 [B][nx]   the new initial state of each trajectory (Solver::setXinit(State))
 *   d_mode  : i32 [B] or NULL (= all 1):  0 leave x0 alone; 1 Solver::initializeWarmstart(state, true)
 *             (acados_solver_interface.cpp:344-364); 2 initializeWarmstart(state, false) (:365-375);
 *             3 Solver::initializeWithBraking(state) (:303-342) with |deceleration| (CONFIG deceleration_at_infeasible)
 *   d_src   : i32 [B] or NULL (= identity): trajectory whose previous solution is shifted into b.
 * Mode 1 writes 0 into the inputs of node 0, where the reference reads State::get(<input>) out of bounds (state.cpp:21-24).
 * The warm start / xinit buffers of the current batch are rewritten in place: the handle's own copies after tmpc_set_batch,
 * the caller's device buffers after tmpc_set_batch_device (they must be writable). */
int tmpc_warmstart(tmpc_handle *h, const void *d_state, const void *d_mode, const void *d_src, double deceleration);
/* GuidanceConstraints::initializeSolverWithGuidance (guidance_constraints.cpp:390-414) for every enabled trajectory:
 * d_gpos, d_gvel f64 [B][N+1][2] (guidance position / velocity at t = k dt), d_enabled u8 [B] or NULL. */
int tmpc_init_with_guidance(tmpc_handle *h, const void *d_gpos, const void *d_gvel, const void *d_enabled);

/* ---- test/debug entry points (used by tests/ to diff per-phase tensors against the oracle) -------- */
/* Copy the batch's (possibly device-built) warm start and xinit back: x0[B][(N+1)*nvar], xinit[B][nx]; either may be NULL. */
int tmpc_debug_get_x0(tmpc_handle *h, double *x0, double *xinit);
/* Evaluate the stage functions on device for n points: z[n][7], p[n][npar] (host pointers).
 * Outputs (host, may be NULL): cost[n], cost_grad[n][7], cost_hess[n][49], h[n][nh], h_jac[n][nh][7],
 * x_next[n][5], x_jac[n][5][7]; lag_hess[n][49] = dt*hess(l) + sum_j pi[j] hess(x_next_j) +
 * sum_r lamh[r] hess(h_r) (pi[n][5], lamh[n][nh] host inputs, NULL = zeros); mirror[n][49] = MIRROR(lag_hess). */
/* Copy the batch's (possibly device-modified) parameter tensor back: params[B][N*npar] host pointer. */
int tmpc_debug_get_params(tmpc_handle *h, double *params);

/* Mean shader-clock cycles per phase over the batch (one extra instrumented solve).  cycles[10]:
 * linearise, residuals, barrier Hessian, Riccati factor, rhs build, Riccati solve, row passes, update, final, total. */
int tmpc_debug_profile(tmpc_handle *h, int64_t *cycles, int32_t n_phases);

/* The LDS bank-conflict model behind the compact kernels' layout padding (no handle, no GPU): the passes the row passes' coefficient loads
 * take per wave and interior-point row pass for a stage stride of `dstride` doubles -- N stages, nh general rows per stage of which the first
 * n_pair are stored as pairs, a kernel of `threads` (64 or 128) threads per trajectory.  tmpc_create picks, among the strides that keep the
 * kernel's residency, the one this function likes best (diagnostics: TMPC_EXP_DPAD in INTEGRATION.md section 7); results never depend on it. */
int tmpc_debug_lds_passes(int32_t N, int32_t n_pair, int32_t nh, int32_t threads, int32_t dstride);
/* Test aid (no reference counterpart): fills the LDS of every CU of the handle's device with signalling-NaN bit patterns (one 160 KB workgroup
 * per CU, several rounds) and waits.  LDS keeps its contents between kernels, so a solve kernel that reads a word it never wrote gives results
 * that depend on what ran before it -- in a fresh test process that is usually zeros and the bug stays invisible (round 6: a column slot the
 * four-wave factorisation multiplied by zero without ever writing it).  tests/test_gpu_lds_poison.py solves after this call and demands the
 * un-poisoned results bit for bit, for every kernel family. */
int tmpc_debug_poison_lds(tmpc_handle *h);
/* 1 if this build of the library reads the TMPC_* lab switches from the environment (libtmpc_hip_lab.so, -DTMPC_LAB_SWITCHES), 0 for the product library. */
int tmpc_has_lab_switches(void);

int tmpc_debug_eval_stage(tmpc_handle *h, int32_t n, const double *z, const double *p, const double *pi,
                          const double *lamh, double *cost, double *cost_grad, double *cost_hess,
                          double *hval, double *h_jac, double *x_next, double *x_jac, double *lag_hess,
                          double *mirror);

#ifdef __cplusplus
}
#endif
#endif
