// The UN-PATCHED drop-in under the reference's threading contract (SURVEY 8b, round-5 verdict next-5): GuidanceConstraints::optimize as the
// reference runs it -- `#pragma omp parallel for num_threads(8)` over the local planners, each doing `*solver = *_solver; ...; loadWarmstart();
// solve()` on its OWN Solver (own tmpc handle, own stream), no locks (mpc_planner_modules/src/guidance_constraints.cpp:279-361) -- against the
// same loop run serially and against the patched module (ONE Solver::solveBatch launch).  All three must agree bit for bit; the tick times of
// the OpenMP loop and of solveBatch are printed for INTEGRATION.md section 4.
//   test_omp_solvers <config dir> <scene.bin> [reps]           (scene format: tests/cpp/test_optimize.cpp)
#include <mpc_planner_modules/modules_hip.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

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

struct Tick
{
    std::shared_ptr<Solver> solver;
    std::unique_ptr<GuidanceConstraints> module;
};

struct Snapshot
{
    int exit_code, best;
    std::vector<int> codes; std::vector<double> objectives, traj;
    bool operator==(const Snapshot &o) const
    {
        return exit_code == o.exit_code && best == o.best && codes == o.codes && objectives.size() == o.objectives.size() && traj.size() == o.traj.size() &&
               std::memcmp(objectives.data(), o.objectives.data(), 8 * objectives.size()) == 0 && std::memcmp(traj.data(), o.traj.data(), 8 * traj.size()) == 0;
    }
};

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    setSolverConfigPath(argv[1]);
    const std::vector<double> in = read_all(argv[2]);
    const int reps = argc > 3 ? std::atoi(argv[3]) : 30;
    size_t o = 0;
    auto next = [&]() { return in[o++]; };
    const int N = (int)next(), M = (int)next(), B = (int)next(), S = (int)next(), tmpcpp = (int)next(), gaussian = (int)next();
    if (gaussian != 0 || N != SOLVER_N || M != SOLVER_MAX_OBSTACLES || S != SOLVER_S) { std::printf("scene does not match the generated solver\n"); return 2; }
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
        ob.prediction = Prediction(PredictionType::DETERMINISTIC);
        for (int i = 0; i < N; i++) { const double x = next(), y = next(); ob.prediction.modes[0].emplace_back(Vector2d(x, y), 0., 0., 0.); }
        ob.position = ob.prediction.modes[0][0].position;
        data.dynamic_obstacles.push_back(ob);
    }
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
    const int selected_before = (int)next();
    if (selected_before >= 0) guidance[selected_before].previously_selected = true;

    // three independent module instances over the same tick (each with its own main solver and local planners = its own tmpc handles)
    auto make_tick = [&]() {
        Tick t;
        t.solver = std::make_shared<Solver>(0);
        t.solver->setXinit(state);
        t.solver->_config["deceleration_at_infeasible"] = 0.0;
        t.solver->initializeWithBraking(state);
        MPCBaseModule base(t.solver, cfg, {"acceleration", "angular_velocity", "velocity", "reference_velocity"});
        Contouring contouring(t.solver, cfg);
        contouring.update(state, data, module_data);
        for (int k = 0; k < N; k++) { base.setParameters(data, module_data, k); contouring.setParameters(data, module_data, k); }
        t.module.reset(new GuidanceConstraints(t.solver, cfg));
        t.module->setGuidanceTrajectories(guidance);
        return t;
    };
    auto snapshot = [&](Tick &t, int exit_code) {
        Snapshot s;
        s.exit_code = exit_code; s.best = t.module->best_planner_index_;
        for (auto &pl : t.module->planners_) { s.codes.push_back(pl.result.exit_code); s.objectives.push_back(pl.result.objective); }
        for (auto &pl : t.module->planners_)                                        // every planner's trajectory, not only the winner's
            for (int k = 0; k <= N; k++)
                for (const char *nm : {"x", "y", "psi", "v", "spline"}) s.traj.push_back(pl.disabled ? 0.0 : pl.local_solver->getOutput(k, nm));
        return s;
    };
    // A fresh tick for every path: fresh capsules (zero multipliers), so the three see identical solver state
    Tick t_serial = make_tick(), t_omp = make_tick(), t_batch = make_tick();
    const Snapshot s_serial = snapshot(t_serial, t_serial.module->optimizeOpenMP(state, data, module_data, 1));           // the same loop on ONE thread
    const Snapshot s_omp = snapshot(t_omp, t_omp.module->optimizeOpenMP(state, data, module_data, 8));                    // guidance_constraints.cpp:279
    const Snapshot s_batch = snapshot(t_batch, t_batch.module->optimize(state, data, module_data));                       // the patched module: one launch
    std::printf("planners %d threads_available %d\n", (int)t_omp.module->planners_.size(), omp_get_max_threads());
    std::printf("exit_code %d best %d successes %d\n", s_omp.exit_code, s_omp.best, (int)std::count(s_omp.codes.begin(), s_omp.codes.end(), 1));
    std::printf("omp_vs_serial_bitwise %d\n", (int)(s_omp == s_serial));
    std::printf("omp_vs_batch_bitwise %d\n", (int)(s_omp == s_batch));
    if (!(s_omp == s_serial) || !(s_omp == s_batch)) {
        for (size_t i = 0; i < s_omp.codes.size(); i++)
            std::printf("planner %zu codes %d %d %d objectives %.17g %.17g %.17g\n", i, s_serial.codes[i], s_omp.codes[i], s_batch.codes[i], s_serial.objectives[i],
                        s_omp.objectives[i], s_batch.objectives[i]);
    }
    // ---- repeated ticks: the capsules keep their multipliers between solves (DESIGN U11), the same in both paths -> results stay equal tick after tick;
    // time per tick = the whole optimize() call (parameter writes, upload, solve, download, selection) ----
    std::vector<double> ms_omp, ms_batch;
    bool same = true;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        const int e1 = t_omp.module->optimizeOpenMP(state, data, module_data, 8);
        auto t1 = std::chrono::steady_clock::now();
        const int e2 = t_batch.module->optimize(state, data, module_data);
        auto t2 = std::chrono::steady_clock::now();
        ms_omp.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        ms_batch.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
        same = same && (snapshot(t_omp, e1) == snapshot(t_batch, e2));
    }
    std::printf("repeated_ticks %d bitwise_equal %d\n", reps, (int)same);
    if (reps > 0) {
        std::sort(ms_omp.begin(), ms_omp.end()); std::sort(ms_batch.begin(), ms_batch.end());
        std::printf("tick_ms_p50 openmp_8_threads %.4f solve_batch %.4f\n", ms_omp[reps / 2], ms_batch[reps / 2]);
        std::printf("tick_ms_p90 openmp_8_threads %.4f solve_batch %.4f\n", ms_omp[std::min(reps - 1, reps * 9 / 10)], ms_batch[std::min(reps - 1, reps * 9 / 10)]);
    }
    return (s_omp == s_serial && s_omp == s_batch && same) ? 0 : 1;
}
