// C++ mirror on a SLACK-model configuration (configuration_safe_horizon, generate_jackalsimulator_solver.py:67-90: model_map.yaml
// has `slack: [x, 7, 0, 5000]`, one entry more than tmpc_dims::lb/ub hold).  Round-1 advisor finding: the model-bound loop wrote
// lb[7]/ub[7] out of range and tmpc_create then failed.  Plumbing on CPU; with `--solve` (GPU) a straight-road SH-MPC problem
// with far-away scenario halfspaces must solve and report the pinned slack state.
#include <mpc_planner_solver/solver_interface.h>
#include <mpc_planner_solver/mpc_planner_parameters.h>

#include <cstdio>
#include <cstring>
#include <string>

using namespace MPCPlanner;
#define ASSERT_TRUE(c) do { if (!(c)) { std::printf("ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    setSolverConfigPath(argc > 1 ? argv[1] : "config");
    const bool do_solve = argc > 2 && std::strcmp(argv[2], "--solve") == 0;
    Solver s(0);
    ASSERT_TRUE(s.nx == 6 && s.nu == 2 && s.nvar == 8 && SOLVER_SLACK == 1);
    ASSERT_TRUE(s._model_map.count("slack") == 1 && s._model_map.at("slack").index == 7);
    State st; st.set("v", 1.5); st.set("slack", 0.0);
    s.setXinit(st);
    s.initializeWithState(st);
    std::printf("plumbing ok\n");
    if (!do_solve) return 0;
    for (int k = 0; k <= s.N; k++) { s.setEgoPrediction(k, "x", 0.3 * k); s.setEgoPrediction(k, "spline", 0.3 * k); s.setEgoPrediction(k, "v", 1.5); }
    const char *w[] = {"acceleration", "angular_velocity", "slack", "velocity", "reference_velocity", "contour", "lag"};
    const double wv[] = {0.34, 0.85, 10000.0, 0.55, 2.0, 0.05, 0.75};
    for (int k = 0; k < s.N; k++) {
        for (int i = 0; i < 7; i++) s.setParameter(k, std::string(w[i]), wv[i]);
        for (int i = 0; i < SOLVER_S; i++) { setSolverParameterSplineXC(k, s._params, 1.0, i); setSolverParameterSplineXD(k, s._params, 6.0 * i, i); setSolverParameterSplineStart(k, s._params, 6.0 * i, i); }
        for (int j = 0; j < SOLVER_NSLK; j++) {                                  // x <= 100: inactive halfspaces (the reference's dummies)
            s.setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_a1", 1.0);
            s.setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_a2", 0.0);
            s.setParameter(k, "disc_0_scenario_constraint_" + std::to_string(j) + "_b", 100.0);
        }
    }
    s.loadWarmstart();
    ASSERT_TRUE(s.solve() == 1);
    ASSERT_TRUE(s.getOutput(5, "v") > 1.5 && s.getOutput(s.N, "x") > 4.0 && s.getOutput(3, "slack") == 0.0);
    std::printf("solve ok: pobj %.6f\n", s._info.pobj);
    return 0;
}
