// C++ end-to-end of the accelerated path: GuidanceConstraints::optimize restated with ONE batched launch
// (mpc_planner_modules/modules_hip.h, the INTEGRATION.md section 4 patch as compiled code) on a scene written by
// tests/test_cpp_optimize.py; prints every planner's result and the selected trajectory for comparison with the Python path.
//   test_optimize <config dir> <scene.bin>
// Scene header: N M B S tmpcpp gaussian; gaussian = 1: Gaussian predictions (per obstacle N x (x, y) then N x (major, minor)), the risk and the
// configured obstacle radius -- for a solver generated with gaussian=True (GaussianConstraints as GUIDANCE_CONSTRAINTS_TYPE).
#include <mpc_planner_modules/modules_hip.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace MPCPlanner;

static std::vector<double> read_all(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::printf("cannot open %s\n", path); std::exit(2); }
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<double> v(n / 8);
    if (std::fread(v.data(), 8, v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    setSolverConfigPath(argv[1]);
    const std::vector<double> in = read_all(argv[2]);
    size_t o = 0;
    auto next = [&]() { return in[o++]; };
    const int N = (int)next(), M = (int)next(), B = (int)next(), S = (int)next(), tmpcpp = (int)next(), gaussian = (int)next();
    if (gaussian != SOLVER_ROW_MODEL) { std::printf("scene does not match the generated solver's row model\n"); return 2; }
    if (N != SOLVER_N || M != SOLVER_MAX_OBSTACLES || S != SOLVER_S) { std::printf("scene does not match the generated solver\n"); return 2; }
    ModuleConfig cfg;
    cfg.max_obstacles = M; cfg.num_segments = S; cfg.n_paths = B; cfg.use_tmpcpp = tmpcpp != 0;
    const char *wn[] = {"acceleration", "angular_velocity", "velocity", "reference_velocity", "contour", "lag", "terminal_angle", "terminal_contouring"};
    for (int i = 0; i < 8; i++) cfg.weights[wn[i]] = next();
    cfg.robot_radius = next();
    const double obstacle_radius = next();
    State state;
    const char *sn[] = {"x", "y", "psi", "v", "spline"};
    for (int i = 0; i < 5; i++) state.set(sn[i], next());
    RealTimeData data;
    data.robot_area.emplace_back(0., cfg.robot_radius);
    for (int j = 0; j < M; j++) {
        DynamicObstacle ob(j, Vector2d(0., 0.), 0., obstacle_radius);
        ob.prediction = Prediction(gaussian ? PredictionType::GAUSSIAN : PredictionType::DETERMINISTIC);
        for (int i = 0; i < N; i++) { const double x = next(), y = next(); ob.prediction.modes[0].emplace_back(Vector2d(x, y), 0., 0., 0.); }
        if (gaussian) for (int i = 0; i < N; i++) { ob.prediction.modes[0][i].major_radius = next(); ob.prediction.modes[0][i].minor_radius = next(); }
        ob.position = ob.prediction.modes[0][0].position;
        data.dynamic_obstacles.push_back(ob);
    }
    if (gaussian) { cfg.risk = next(); cfg.obstacle_radius = obstacle_radius; }
    ModuleData module_data;
    for (int i = 0; i < S; i++) {
        PathSegment sg;
        sg.ax = next(); sg.bx = next(); sg.cx = next(); sg.dx = next(); sg.ay = next(); sg.by = next(); sg.cy = next(); sg.dy = next(); sg.start = next();
        module_data.path.push_back(sg);
    }
    std::vector<GuidanceTrajectory> guidance(B);
    for (int b = 0; b < B; b++) {
        guidance[b].topology_class = b;
        for (int k = 0; k <= N; k++) { const double x = next(), y = next(); guidance[b].positions.emplace_back(x, y); }
        for (int k = 0; k <= N; k++) { const double x = next(), y = next(); guidance[b].velocities.emplace_back(x, y); }
    }
    const int selected_before = (int)next();                                    // previously selected topology (-1: none)
    if (selected_before >= 0) guidance[selected_before].previously_selected = true;

    // main solver: Planner::solveMPC's preparation (planner.cpp:64-113): xinit, forward-propagated warm start, objective modules
    auto solver = std::make_shared<Solver>(0);
    solver->setXinit(state);
    solver->_config["deceleration_at_infeasible"] = 0.0;       // initializeWithBraking with a = 0: the forward-propagated warm start of
    solver->initializeWithBraking(state);                       // scenes.py (modules.initialize_with_forward_propagation), :303-342
    MPCBaseModule base(solver, cfg, {"acceleration", "angular_velocity", "velocity", "reference_velocity"});
    Contouring contouring(solver, cfg);
    contouring.update(state, data, module_data);
    for (int k = 0; k < N; k++) { base.setParameters(data, module_data, k); contouring.setParameters(data, module_data, k); }

    GuidanceConstraints guidance_constraints(solver, cfg);
    guidance_constraints.setGuidanceTrajectories(guidance);
    const int exit_code = guidance_constraints.optimize(state, data, module_data);
    std::printf("exit_code %d best %d\n", exit_code, guidance_constraints.best_planner_index_);
    for (auto &pl : guidance_constraints.planners_)
        std::printf("planner %d disabled %d exit %d objective %.17g guidance_id %d\n", pl.id, (int)pl.disabled, pl.result.exit_code, pl.result.objective, pl.result.guidance_ID);
    for (int k = 0; k <= N; k++)
        std::printf("x %d %.17g %.17g %.17g %.17g %.17g\n", k, solver->getOutput(k, "x"), solver->getOutput(k, "y"), solver->getOutput(k, "psi"),
                    solver->getOutput(k, "v"), solver->getOutput(k, "spline"));
    // next tick's bookkeeping (:106-108, 192-250): the same trajectories arrive in reverse order plus one new class -- every known class finds
    // the planner that solved it last tick (existing_guidance: the `warmstart_with_mpc_solution` branch, :310, is reachable)
    {
        std::vector<GuidanceTrajectory> next_tick(guidance.rbegin(), guidance.rend());
        next_tick.front().topology_class = 1000;                             // a class nobody solved
        guidance_constraints.setGuidanceTrajectories(next_tick);
        for (auto &kv : guidance_constraints.guidanceToPlannerMap()) std::printf("map %d %d\n", kv.first, kv.second);
        for (auto &pl : guidance_constraints.planners_) std::printf("taken %d %d %d\n", pl.id, (int)pl.taken, (int)pl.existing_guidance);
    }
    // the parameter rows the best planner was solved with (a3-a5 on the C++ side)
    for (int k = 0; k < N; k++) {
        std::printf("p %d", k);
        for (int i = 0; i < SOLVER_NP; i++) std::printf(" %.17g", solver->_params.all_parameters[k * SOLVER_NP + i]);
        std::printf("\n");
    }
    return 0;
}
