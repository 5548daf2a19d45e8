/*
 * oracle/tmpc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C FP64 restatement of what tud-amr/mpc_planner's `Solver::solve()` makes acados do for the
 * Jackal T-MPC problem (SURVEY.md Appendix A + B).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may link or call this library; the product path
 * (mpc_planner_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" at the acados boundary -- acados/HPIPM/CasADi are un-vendored,
 * unpinned third parties absent from /root/reference (README.md:225-232) and the reference's own tests
 * assert no numerical solve result.  What IS pinned:
 *   - stage cost / constraints / dynamics and all first+second derivatives against golden vectors made
 *     by executing the reference's own python modules (tests/golden/make_golden.py),
 *   - the reference-test anchors objective = 25.0, constraint = 125.0 (test_control_modules.py:56-59,89-95),
 *   - parameter counts / index map (test_control_modules.py:53,86),
 *   - converged-SQP optimum against scipy SLSQP on the same NLP (tests/test_oracle_solve.py).
 * Every [UPSTREAM] default assumed about acados is listed in DESIGN.md.
 *
 * Reference files restated (see each function for file:line):
 *   solver_generator/solver_model.py:193-214          dynamics, bounds
 *   solver_generator/generate_acados_solver.py:86-177  OCP + solver options
 *   solver_generator/spline.py:4-86, util/math.py:5-7  glued spline, rotation matrix
 *   mpc_planner_modules/scripts/{mpc_base,contouring,ellipsoid_constraints,guidance_constraints}.py
 *   mpc_planner_solver/src/acados_solver_interface.cpp:86-204,274-284   runtime protocol
 *   mpc_planner_modules/src/guidance_constraints.cpp:264-434            batch loop + selection
 */
#ifndef TMPC_ORACLE_H
#define TMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Two builds of the same sources (oracle/Makefile):
 *   liboracle_tmpc.so        ORC_SLACK=0  ContouringSecondOrderUnicycleModel          (solver_model.py:193-214)
 *   liboracle_tmpc_slack.so  ORC_SLACK=1  ContouringSecondOrderUnicycleModelWithSlack (solver_model.py:274-298):
 *                                         one more state `slack` with slack' = 0, appended after `spline`.
 * ORC_NXE / ORC_NVE are the MODEL's dimensions (array strides of xinit, x0, xtraj and of every stage-function
 * derivative); ORC_NX / ORC_NV are what the QP carries -- the slack state is presolved out of the QP (sqp_rti.c, U9). */
#ifndef ORC_SLACK
#define ORC_SLACK 0
#endif
#define ORC_NU 2
#define ORC_NX 5
#define ORC_NV 7
#define ORC_NXE (ORC_NX + ORC_SLACK)
#define ORC_NVE (ORC_NV + ORC_SLACK)
#define ORC_MAX_N 32
#define ORC_MAX_NH 40   /* general inequality rows per stage */

/* Problem description: sizes + parameter index map (util/parameters.py:25-55). */
typedef struct {
    int N;          /* horizon (shooting intervals); nodes 0..N */
    int S;          /* contouring spline segments */
    int n_lin;      /* topology halfspace rows per stage (0 = no GuidanceConstraints module) */
    int M;          /* ellipsoid rows per stage (max_obstacles * n_discs, n_discs = 1) */
    int npar;       /* parameters per stage */
    double dt;      /* integrator_step */
    /* solver options (generate_acados_solver.py:143-177, settings.yaml:16-17) */
    int n_sqp;          /* RTI iterations per solve() */
    int qp_iter_max;    /* 50 */
    double qp_tol;      /* 1e-5 */
    double reg_eps;     /* MIRROR epsilon, [UPSTREAM default] 1e-4 */
    double ipm_mu0;     /* IPM initial barrier */
    double ipm_thr0;    /* IPM slack floor at initialisation */
    int erk_steps;      /* 3 */
    /* state/input box bounds, order [a,w,x,y,psi,v,spline] (solver_model.py:204-205) */
    double lb[ORC_NV];
    double ub[ORC_NV];
    /* rows a1*x + a2*y - (b + slack) <= 0 of DecompConstraintModule / ScenarioConstraintModule
     * (decomp_constraints.py:68-98, scenario_constraints.py:64-94), after the ellipsoid rows */
    int n_slk;
    int slack;          /* = ORC_SLACK: the model has the slack state (and MPCBase weighs it) */
    double lb_slack, ub_slack;   /* solver_model.py:285-286: [0, 5000] */
    /* Gaussian chance-constraint rows (gaussian_constraints.py:33-113; mpc_planner_jackal's default T-MPC uses them as the
     * guidance submodule instead of the ellipsoids, generate_jackal_solver.py:53-73): rows >= 0, after the topology rows.
     * Only with M == 0 and n_slk == 0. */
    int n_gauss;
    /* fraction-to-the-boundary factor of the interior-point step.  An ORACLE option on purpose: tests re-run the oracle with
     * textbook interior-point constants (0.995, mu0 = thr0 = 1) to show that its RTI iterate -- which tests/independent_rti.py
     * pins with an active-set QP solver -- does not depend on the constants the kernels were tuned with (0.999, 0.01, 0.01). */
    double ipm_tau;
    /* 0: ContouringModule (contouring.py:48-98: lag / contour errors along the path's tangent frame);
     * 1: CurvatureAwareContouringModule (curvature_aware_contouring.py:48-105: squared distance to the path point + tracking of the
     *    projected path velocity s_dot) -- BASELINE configs[2] "CA-MPC".  Same parameter map (its C++ side leaves velocity /
     *    reference_velocity to MPCBaseModule, curvature_aware_contouring.cpp:15-49), same model (s' = v, SURVEY Appendix D-8 route 1). */
    int cost_model;
    /* ---- interior-point options that bring the oracle's QP solver closer to what the reference configures (round-3 verdict item 5;
     * HPIPM itself is absent, so these follow its published behaviour -- "HPIPM-like", not HPIPM).  Oracle options only: the kernels
     * run the tuned cold start (all zero here). ----
     * qp_warm_start: the reference sets qp_solver_warm_start = 2 (generate_acados_solver.py:173).  0: every QP of a solve starts cold
     *   (tuned default); 1: primal variables of the previous QP of the same solve; 2: primal AND dual variables (pi, lam, t) of the
     *   previous QP, as HPIPM's warm_start = 2 takes them (floored at 1e-6, cf. its t_min / lam_min).  The first QP of a solve is always cold: the reference
     *   resets the QP memory with every `*solver = *_solver` and after a failed solve (acados_solver_interface.cpp:70,190).
     * ipm_init_box: HPIPM's cold start moves the primal start into the interior of the box constraints by thr0 before the slacks are
     *   initialised (d_ocp_qp_init_var: ux = lb + thr0 / ub - thr0 / the middle), instead of starting at dz = 0 whatever the boxes say. */
    int qp_warm_start;
    int ipm_init_box;
    /* riccati_form: how the interior-point method's Riccati recursion carries the cost-to-go Hessian P_k from stage to stage.
     *   0 (default): square-root form -- the 7x7 stage matrix is Cholesky-factorised completely, P_k = Lxx Lxx^T, F = Hh + (Lxx^T [B A])^T (Lxx^T [B A])
     *      -- what HPIPM's default (mode BALANCE, square_root_alg = 1) does [UPSTREAM];
     *   1: the same elimination stopped after the two input columns: P_k = F_xx - Lxu Lxu^T is used as it is, F = Hh + [B A]^T P [B A]
     *      (HPIPM's square_root_alg = 0, its SPEED modes) -- what the HIP kernels run since round 5 (csrc/tmpc_riccati.hpp).
     * The first two column eliminations are the same operations in both forms.  tools/riccati_form_study.py: at the reference's qp_tol = 1e-5
     * the two forms give the same exit codes and iteration counts on every bench scene and iterates equal to 4e-11; at qp_tol = 1e-9, where the
     * interior-point method runs into the conditioning of the barrier systems, 0.3 % (cfg 2) to 5 % (cfg 3) of the solves end differently --
     * tests that compare the kernels with the oracle at such tolerances set this option so that both sides run the same algorithm. */
    int riccati_form;
    /* model: 0 ContouringSecondOrderUnicycleModel (solver_model.py:193-214: spline' = v); 1 SecondOrderUnicycleModel (solver_model.py:170-191:
     * states x, y, psi, v).  The oracle keeps its 5-state arrays for model 1 with the fifth slot inert (its ODE is s' = 0 and no cost or row reads
     * it): it decouples exactly from the four real states, which is how the HIP kernels run the model too (csrc/tmpc_stage.hpp Dims::model).
     * Set by orc_problem_set_goal_stack together with cost_model 2 = MPCBase weights on (a, w, v) + GoalObjective (goal_module.py:22-36). */
    int model;
} orc_problem;

/* HPIPM-like interior-point settings on a problem (mode BALANCE as published in hpipm's d_ocp_qp_ipm_arg_set_default: mu0 = 1e1,
 * alpha_min = 1e-12, Mehrotra predictor-corrector with sigma = (mu_aff / mu)^3, one step length for primal and dual, fraction to the
 * boundary 0.995; d_ocp_qp_init_var: thr0 = 0.1, primal start inside the boxes; acados passes qp_tol as all four res_*_max and
 * iter_max = 50).  warm_start: 0 or the reference's 2. */
void orc_problem_set_hpipm_like(orc_problem *pb, int warm_start);
/* The goal-tracking stack on SecondOrderUnicycleModel (model 1) or on the contouring model (model 0): MPCBaseModule(a, w, v) + GoalModule +
 * EllipsoidConstraintModule(M obstacles, as given to orc_problem_init*): parameters acceleration, angular_velocity, velocity, reference_velocity,
 * goal_weight, goal_x, goal_y, ego_disc_radius, ego_disc_0_offset, 7 per obstacle; no spline segments, no topology rows.  Model 1 also takes the
 * model's own bounds (solver_model.py:179-180). */
void orc_problem_set_goal_stack(orc_problem *pb, int model);

/* Fill sizes, default options and bounds for the Jackal contouring unicycle. */
void orc_problem_init(orc_problem *pb, int N, int S, int n_lin, int M);
void orc_problem_init_ex(orc_problem *pb, int N, int S, int n_lin, int M, int n_slk);
int orc_model_nx(void);   /* ORC_NXE of this build */
/* Replace the (absent) ellipsoid module by n_gauss Gaussian chance-constraint rows; recomputes npar. */
void orc_problem_set_gaussian(orc_problem *pb, int n_gauss);
int orc_idx_gaussian(const orc_problem *pb, int j, int which);  /* which: 0..5 = x,y,major,minor,risk,r */

/* parameter index helpers (index into one stage's parameter row) */
int orc_idx_weight(const orc_problem *pb, int which);           /* 0..7: acceleration, angular_velocity, velocity,
                                                                   reference_velocity, contour, lag, terminal_angle,
                                                                   terminal_contouring; 8: slack (slack build) */
int orc_idx_spline(const orc_problem *pb, int seg, int which);  /* which: 0..8 = xa,xb,xc,xd,ya,yb,yc,yd,start */
int orc_idx_lin(const orc_problem *pb, int j, int which);       /* which: 0..2 = a1,a2,b */
int orc_idx_disc_radius(const orc_problem *pb);
int orc_idx_disc_offset(const orc_problem *pb);
int orc_idx_ellipsoid(const orc_problem *pb, int j, int which); /* which: 0..6 = x,y,psi,major,minor,chi,r */
int orc_idx_slk(const orc_problem *pb, int j, int which);       /* which: 0..2 = a1,a2,b */

/* ---- stage functions with exact first/second derivatives (forward-mode 2nd-order jets) ---------- */
/* z = [a,w,x,y,psi,v,spline(,slack)] (ORC_NVE entries); p = one stage's parameter row. */
void orc_stage_cost(const orc_problem *pb, const double *z, const double *p,
                    double *val, double grad[ORC_NVE], double hess[ORC_NVE * ORC_NVE]);
/* h[0..n_lin) topology rows (<= 0), h[n_lin..n_lin+M) ellipsoid rows (>= 1) or n_gauss chance rows (>= 0), then n_slk rows (<= 0) */
void orc_stage_constraints(const orc_problem *pb, const double *z, const double *p,
                           double *h, double *jac /* nh x NVE */, double *hess /* nh x NVE x NVE */);
void orc_continuous_dynamics(const double *z, double f[ORC_NXE]);
/* ERK4 x erk_steps over dt.  jac: NXE x NVE (d x_next / d [u;x]); hess: NXE x NVE x NVE */
void orc_discrete_dynamics(const orc_problem *pb, const double *z,
                           double xnext[ORC_NXE], double *jac, double *hess);
void orc_constraint_bounds(const orc_problem *pb, double *lh, double *uh);

/* MIRROR regularisation of an n x n symmetric matrix (row-major, in place): V max(|e|,eps) V^T */
void orc_mirror(double *W, int n, double eps);

/* ---- one solve ------------------------------------------------------------------------------- */
typedef struct {
    double pobj;      /* ocp_nlp_eval_cost at the returned iterate */
    double res_eq;    /* inf-norm of dynamics defects + initial condition at the returned iterate */
    int exit_code;    /* Forces-style: 1 success, 0 generic failure, 2 maxit, 3 minstep, 4 QP failure */
    int qp_status;    /* 0 ok, 2 max iter, 3 min step, 4 NaN */
    int sqp_iter;     /* RTI iterations executed */
    int qp_iter_total;/* IPM iterations summed over all QPs */
} orc_info;

/* xinit[NXE]; x0[(N+1)*NVE] warm start laid out [u_k; x_k] per node (acados_solver_interface.h:53-54);
 * params[N*npar]; xtraj[(N+1)*NXE]; utraj[N*NU].  Fresh solver state (zero multipliers). */
void orc_solve(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
               double *xtraj, double *utraj, orc_info *info);

/* Same with multipliers carried in / out (the reference's capsules keep them between solves and reset them after a failed one:
 * acados_solver_interface.cpp:67-77,187-191,274-284): pi_io [(N+1) NX], lamh_io [N ORC_MAX_NH]; n_iter RTI iterations. */
void orc_solve_carry(const orc_problem *pb, const double *xinit, const double *x0, const double *params, int n_iter,
                     double *pi_io, double *lamh_io, double *xtraj, double *utraj, orc_info *info);

/* Optional debug capture of the first linearisation / QP of a solve (for per-phase GPU diffing). */
typedef struct {
    double W[(ORC_MAX_N + 1) * ORC_NV * ORC_NV];
    double g[(ORC_MAX_N + 1) * ORC_NV];
    double BA[ORC_MAX_N * ORC_NX * ORC_NV];
    double b[ORC_MAX_N * ORC_NX];
    double h[ORC_MAX_N * ORC_MAX_NH];
    double D[ORC_MAX_N * ORC_MAX_NH * ORC_NV];
    double W_raw[(ORC_MAX_N + 1) * ORC_NV * ORC_NV];  /* Lagrangian Hessian before MIRROR */
    double z_in[(ORC_MAX_N + 1) * ORC_NV];             /* linearisation point */
    double pi_in[(ORC_MAX_N + 1) * ORC_NX];            /* multipliers used in W_raw */
    double lamh_in[ORC_MAX_N * ORC_MAX_NH];
    double dz[(ORC_MAX_N + 1) * ORC_NV];   /* QP solution */
    double pi[(ORC_MAX_N + 1) * ORC_NX];
    int qp_iters;
} orc_debug;
void orc_solve_debug(const orc_problem *pb, const double *xinit, const double *x0, const double *params,
                     double *xtraj, double *utraj, orc_info *info, orc_debug *dbg, int capture_sqp_iter);

/* ---- batch (restates the OpenMP loop of guidance_constraints.cpp:279-361) ------------------------ */
/* B trajectories; xinit[B][NXE], x0[B][(N+1)*NVE], params[B][N*npar]; outputs per trajectory. */
void orc_solve_batch(const orc_problem *pb, int B, const double *xinit, const double *x0,
                     const double *params, double *xtraj, double *utraj, orc_info *info, int num_threads);

/* Same, recording the interior-point iterations of every QP in qp_iters[B][n_sqp] (workload statistics for DESIGN.md). */
void orc_solve_batch_trace(const orc_problem *pb, int B, const double *xinit, const double *x0,
                           const double *params, double *xtraj, double *utraj, orc_info *info, int num_threads, int *qp_iters);

/* FindBestPlanner (guidance_constraints.cpp:416-434): argmin of objective*weight over enabled &
 * success; init 1e10, strict '<' => lowest index wins ties; -1 if none. */
int orc_find_best(int B, const double *objective, const int *exit_code, const unsigned char *disabled);

#ifdef __cplusplus
}
#endif
#endif
