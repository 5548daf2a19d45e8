/*
 * mpc_planner_solver/solver_interface.h -- HIP flavour of the reference's solver interface.
 *
 * Source-compatible mirror of `MPCPlanner::Solver` (reference:
 * mpc_planner_solver/include/mpc_planner_solver/acados_solver_interface.h:93-222 and
 * src/acados_solver_interface.cpp) so that the modules (mpc_planner_modules/src/*.cpp) and
 * Planner::solveMPC (mpc_planner/src/planner.cpp:37-158) compile against it unchanged: same public members
 * (_params, _info, _output, N, nu, nx, nvar, npar, dt, _num_iterations, _solver_id), same methods, same
 * exit-code convention.  The acados capsule is replaced by a handle of the batch-first C-ABI
 * (include/tmpc_hip.h); `solveBatch` is the new entry point that lets GuidanceConstraints::optimize
 * (guidance_constraints.cpp:279-361) hand all local planners to ONE launch instead of an OpenMP loop.
 *
 * yaml-cpp / Eigen / ros_tools are not available in the build image, so the three generated YAML maps are read
 * by a ~40-line flat reader (the generated files are flat "key: value" / "key: [a, b, c, d]" maps) and
 * positions use a 2-double struct with Eigen's (i) accessor.
 */
#ifndef MPC_PLANNER_HIP_SOLVER_INTERFACE_H
#define MPC_PLANNER_HIP_SOLVER_INTERFACE_H

#include <map>
#include <string>
#include <vector>

#include <mpc_planner_solver/hip_solver_dims.h>
#include "tmpc_hip.h"

#define NX SOLVER_NX
#define NU SOLVER_NU

namespace MPCPlanner
{
    struct Vector2d
    {
        double v[2];
        Vector2d(double x = 0., double y = 0.) : v{x, y} {}
        double &operator()(int i) { return v[i]; }
        double operator()(int i) const { return v[i]; }
    };

    struct ModelEntry { std::string type; int index; double lb, ub; };          // model_map.yaml row
    typedef std::map<std::string, int> ParameterMap;                            // parameter_map.yaml
    typedef std::map<std::string, ModelEntry> ModelMap;
    void loadSolverConfig(const std::string &dir, std::map<std::string, double> &settings, ParameterMap &pm, ModelMap &mm);
    void setSolverConfigPath(const std::string &dir);                           // where the generated YAMLs live

    /* state.h:13-30 */
    struct State
    {
        State();
        void initialize();
        double get(std::string &&var_name) const;
        Vector2d getPos() const;
        void set(std::string &&var_name, double value);
    private:
        std::vector<double> _state;
        ModelMap _model_map;
        int _nu;
    };

    /* acados_solver_interface.h:51-91 -- trivially copyable host arrays, identical layout */
    struct AcadosParameters
    {
        double xinit[NX];
        double x0[(NU + NX) * (SOLVER_N + 1)];
        double all_parameters[SOLVER_NP * SOLVER_N];
        double solver_timeout{0.};
        double *getU0() { return x0; }
        AcadosParameters();
        void printParameters(const ParameterMap &parameter_map) const;      /* :76-88 (YAML::Node -> the flat map read from parameter_map.yaml) */
    };

    class BatchContext;
    class Solver
    {
    public:
        enum class StageIndexing;
        struct AcadosInfo              /* :96-125 */
        {
            double min_time, kkt_norm_inf, elapsed_time;
            int sqp_iter;
            double nlp_res, solvetime;
            int qp_status;
            double pobj{0.};
            AcadosInfo() : min_time(1e12), kkt_norm_inf(0.), elapsed_time(0.), sqp_iter(0), nlp_res(0.), solvetime(0.), qp_status(0) {}
            void print() const;              /* :115-124 without the capsule argument (no acados statistics to print) */
        };
        struct AcadosOutput            /* :127-148 */
        {
            double xtraj[NX * (SOLVER_N + 1)];
            double utraj[NU * SOLVER_N];
            AcadosOutput();
            void print() const;              /* :141-147 */
        };

    private:
        void recordTiming(double launch_seconds, int iterations);     // _info.elapsed_time / solvetime / min_time (:151-153)
        tmpc_handle *_handle{nullptr};          // replaces the acados capsule; created lazily on the first solve.  Like the capsule
                                                // it keeps the NLP iterate and its multipliers between calls (tmpc_solve_iterations)
        int _exit_code_one_iter{-1};
        int _device{0};
        bool _warmstart_pending{true};          // loadWarmstart() since the last iteration: the next one starts from _params.x0
        bool _new_solve{true};                  // initializeOneIteration() since the last iteration: the next one opens a new solve (TMPC_ITER_NEW_SOLVE)
        int _iterations_done{0};                // RTI iterations since initializeOneIteration
        std::vector<BatchContext *> _contexts;  // contexts this Solver owns a slot of: ~Solver releases the slots (a new Solver at the same
                                                // address must not inherit this one's multipliers)
        double _iteration_time_estimate{0.};
        StageIndexing _stage_indexing{StageIndexing::AcadosNodes};
        void ensureHandle();
        int runIterations(int n, bool complete);

    public:
        int _solver_id;
        AcadosParameters _params;
        AcadosInfo _info;
        AcadosOutput _output;
        int N;
        unsigned int nu, nx, nvar, npar;
        double dt;
        std::map<std::string, double> _config;
        ParameterMap _parameter_map;
        ModelMap _model_map;
        int _num_iterations;

        Solver(int solver_id = 0);
        ~Solver();
        Solver(const Solver &) = delete;
        Solver &operator=(const Solver &rhs);   // copies _params only (acados_solver_interface.cpp:67-77)
        void reset();

        /* The reference leaves its RTI loop when the planning time runs out (:111-116: elapsed + average iteration time >=
         * _params.solver_timeout), which makes the iteration count wall-clock dependent.  Deterministic counterpart: with an
         * iteration-time estimate set (seconds per RTI iteration, > 0) and _params.solver_timeout > 0, solve() runs
         * min(_num_iterations, max(1, floor(solver_timeout / estimate))) iterations; 0 (default) keeps the fixed budget. */
        void setIterationTimeEstimate(double seconds_per_iteration) { _iteration_time_estimate = seconds_per_iteration; }
        int iterationBudget() const;

        int solve();                            // :86-119
        void initializeOneIteration();          // :121-143 xinit + parameters to the device
        int solveOneIteration();                // :145-160 exactly ONE RTI iteration, continuing from the iterate / multipliers the
                                                //          handle holds (or from _params.x0 after loadWarmstart()); returns the acados-style
                                                //          status (0, or 4 after a QP failure) and sets _info.qp_status
        int completeOneIteration();             // :162-204 cost, trajectories, res_eq test, capsule reset on failure, exit-code mapping

        /* GuidanceConstraints::optimize / ScenarioConstraints::optimize batch path: solvers[i]->_params in, _output/_info out, exit codes
         * returned.  The batch belongs to the CALLER (one BatchContext per module instance, below): every Solver that goes through a
         * context owns one state slot of it for the context's lifetime, like the reference's one capsule per Solver (:17,51-65). */
        static std::vector<int> solveBatch(BatchContext &ctx, const std::vector<Solver *> &solvers);
        friend class BatchContext;

        bool hasParameter(std::string &&parameter);
        void setParameter(int k, std::string &&parameter, double value);
        void setParameter(int k, std::string &parameter, double value);
        double getParameter(int k, std::string &&parameter);
        void setXinit(std::string &&state_name, double value);
        void setXinit(const State &state);
        void setEgoPrediction(unsigned int k, std::string &&var_name, double value);
        double getEgoPrediction(unsigned int k, std::string &&var_name);
        void setEgoPredictionPosition(unsigned int k, const Vector2d &value);
        Vector2d getEgoPredictionPosition(unsigned int k);
        void loadWarmstart();
        void initializeWarmstart(const State &state, bool shift_previous_solution_forward);
        void initializeWithState(const State &initial_state);
        void initializeWithBraking(const State &initial_state);
        double getOutput(int k, std::string &&state_name) const;
        /* Forces-style access (forces_solver_interface.cpp:241-244: getForcesOutput(_output, k, index) reads entry `index` of the
         * stage vector z_k = [u_k; x_k] of stage k = 0 .. N-1 -- the Forces flavour has N stages x01 .. xN, the acados flavour N + 1
         * nodes): callers written against the Forces interface index the same solution through this accessor. */
        double getForcesStyleOutput(int k, int index) const { return index < (int)nu ? _output.utraj[k * nu + index] : _output.xtraj[k * nx + index - nu]; }
        /* Forces N-stage indexing as a MODE (forces_solver_interface.cpp:147-182, 241-244).  Callers written against the reference's
         * Forces flavour see N stages z_0 .. z_{N-1} = [u_k; x_k] and nothing else: with the mode on, getOutput(k, name) accepts k in [0, N) only
         * (the node N of this solver's horizon is not a Forces stage: asking for it is a caller error and aborts like an out-of-range Forces
         * output would), stages() is N, and initializeWarmstart follows the Forces loops -- k < N, the terminal stage extrapolated from output
         * N-1 (:159-160) resp. kept (:171-179) -- and fills this solver's extra node N with the same terminal value so that the horizon it
         * actually solves stays defined.  The problem solved is unchanged (N intervals, ERK4 x 3, zero terminal cost: the acados flavour):
         * this is an indexing / warm-start compatibility mode, not the Forces NLP (SURVEY Appendix D-1 lists what that would change). */
        enum class StageIndexing { AcadosNodes, ForcesStages };
        void setStageIndexing(StageIndexing mode) { _stage_indexing = mode; }
        StageIndexing stageIndexing() const { return _stage_indexing; }
        int stages() const { return _stage_indexing == StageIndexing::ForcesStages ? N : N + 1; }
        std::string explainExitFlag(int exitflag) const;
        void printIfBoundLimited() const;
    };

    /* One batched launch for several Solvers, owned by the caller.  Holds the HIP handle of the batch and one persistent state slot
     * (NLP iterate + multipliers, the capsule's state) per Solver that ever went through it: a planner keeps its slot whatever
     * subset of the planners a tick launches (disabled planners are skipped, guidance_constraints.cpp:286-293) and whatever other
     * module instances do with their own contexts.  The handle grows (state copied) when more Solvers appear than it has slots.
     * Not thread safe: like a Solver, a context belongs to one caller. */
    class BatchContext
    {
    public:
        BatchContext() = default;
        ~BatchContext();
        BatchContext(const BatchContext &) = delete;
        BatchContext &operator=(const BatchContext &) = delete;
        std::vector<int> solve(const std::vector<Solver *> &solvers);
        int slotOf(const Solver *s) const { auto it = _slot.find(s); return it == _slot.end() ? -1 : it->second; }
        int capacity() const { return _capacity; }
        void forget(const Solver *s);                                 // a Solver that is destroyed while the context lives (~Solver calls it): its slot is cleared and reused
        double lastLaunchSeconds() const { return _last_launch_s; }

    private:
        void ensure(const Solver *s0, int needed_slots);
        tmpc_handle *_handle{nullptr};
        int _capacity{0}, _device{-1}, _iterations{-1}, _next_slot{0};
        double _dt{0.}, _last_launch_s{0.};
        std::map<const Solver *, int> _slot;
        std::vector<double> _stage_xinit, _stage_x0, _stage_par;      // host staging of a tick's inputs (kept between ticks)
        std::vector<int> _free_slots;           // slots of forgotten Solvers, cleared (tmpc_clear_slot) before they are handed out again
        int takeSlot();
    };
}
#endif
