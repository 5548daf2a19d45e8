// C++ ScenarioConstraints::optimize restated with one batched launch (modules_hip.h; scenario_constraints.cpp:58-108) on P
// parallel scenario solvers whose halfspaces come from a file written by tests/test_cpp_optimize.py.
//   test_scenario_optimize <config dir> <scene.bin>
#include <mpc_planner_modules/modules_hip.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace MPCPlanner;

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    setSolverConfigPath(argv[1]);
    FILE *f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<double> in(n / 8);
    if (std::fread(in.data(), 8, in.size(), f) != in.size()) return 2;
    std::fclose(f);
    size_t o = 0;
    auto next = [&]() { return in[o++]; };
    const int N = (int)next(), P = (int)next(), R = (int)next(), S = (int)next();
    if (N != SOLVER_N || R != SOLVER_NSLK || S != SOLVER_S) { std::printf("scene does not match the generated solver\n"); return 2; }
    ModuleConfig cfg; cfg.num_segments = S;
    const char *wn[] = {"acceleration", "angular_velocity", "slack", "velocity", "reference_velocity", "contour", "lag", "terminal_angle", "terminal_contouring"};
    for (int i = 0; i < 9; i++) cfg.weights[wn[i]] = next();
    State state;
    const char *sn[] = {"x", "y", "psi", "v", "spline", "slack"};
    for (int i = 0; i < 6; i++) state.set(sn[i], next());
    RealTimeData data; ModuleData module_data;
    for (int i = 0; i < S; i++) {
        PathSegment sg;
        sg.ax = next(); sg.bx = next(); sg.cx = next(); sg.dx = next(); sg.ay = next(); sg.by = next(); sg.cy = next(); sg.dy = next(); sg.start = next();
        module_data.path.push_back(sg);
    }
    auto solver = std::make_shared<Solver>(0);
    solver->setXinit(state);
    for (int k = 0; k <= N; k++)                                      // the main solver's warm start as written by the caller
        for (int i = 0; i < 8; i++) solver->_params.x0[k * 8 + i] = next();
    MPCBaseModule base(solver, cfg, {"acceleration", "angular_velocity", "slack", "velocity", "reference_velocity"});
    Contouring contouring(solver, cfg);
    contouring.update(state, data, module_data);
    for (int k = 0; k < N; k++) { base.setParameters(data, module_data, k); contouring.setParameters(data, module_data, k); }
    ScenarioConstraints sc(solver, P, 0.0);
    for (int p = 0; p < P; p++) {
        auto &hs = sc._scenario_solvers[p]->halfspaces;
        hs.resize(N);
        for (int k = 1; k < N; k++)
            for (int r = 0; r < R; r++) { const double a1 = next(), a2 = next(), b = next(); hs[k].emplace_back(Vector2d(a1, a2), b); }
        auto &sr = sc._scenario_solvers[p]->scenario_of_row;          // the scenario behind each row (-1: dummy row)
        sr.resize(N);
        for (int k = 1; k < N; k++)
            for (int r = 0; r < R; r++) sr[k].push_back((int)next());
    }
    sc._support_tolerance = next();
    const int exit_code = sc.optimize(state, data, module_data);
    int best = -1;
    for (int p = 0; p < P; p++) if (sc._best_solver == sc._scenario_solvers[p].get()) best = p;
    std::printf("exit_code %d best %d\n", exit_code, best);
    for (int p = 0; p < P; p++)
        std::printf("solver %d exit %d objective %.17g support %d rows %d\n", p, sc._scenario_solvers[p]->exit_code, sc._scenario_solvers[p]->solver->_info.pobj,
                    sc._scenario_solvers[p]->support.size(), sc._scenario_solvers[p]->support.active_rows);
    for (int k = 0; k <= N; k++) std::printf("x %d %.17g %.17g %.17g %.17g\n", k, solver->getOutput(k, "x"), solver->getOutput(k, "y"), solver->getOutput(k, "psi"), solver->getOutput(k, "v"));
    return 0;
}
