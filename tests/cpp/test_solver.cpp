// C++ test of the HIP-flavour MPCPlanner::Solver, written after the reference's
// mpc_planner_solver/test/test_solver.cpp:52-133 (State set/get; dims; setParameter/getParameter; setXinit sums;
// ego-prediction indices; operator= copies params).  The Forces-only `_output.x01/x02/x03` block (:119-121) is
// replaced by its acados-flavour equivalent (initializeWarmstart from xtraj).  With `--solve` (GPU) it also checks
// solve() against solveBatch() and the exit-code convention.
#include <mpc_planner_solver/solver_interface.h>
#include <mpc_planner_solver/mpc_planner_parameters.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

using namespace MPCPlanner;
#define ASSERT_TRUE(c) do { if (!(c)) { std::printf("ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    setSolverConfigPath(argc > 1 ? argv[1] : "config");
    const bool do_solve = argc > 2 && std::strcmp(argv[2], "--solve") == 0;
    {
        State state;
        state.set("x", 1.5); state.set("y", 3.5);
        ASSERT_TRUE(state.get("y") == 3.5); ASSERT_TRUE(state.get("x") == 1.5);
        ASSERT_TRUE(state.getPos()(0) == 1.5); ASSERT_TRUE(state.getPos()(1) == 3.5);
    }
    Solver solver;
    ASSERT_TRUE(solver.nu + solver.nx == solver.nvar);
    ASSERT_TRUE(solver.npar > 0); ASSERT_TRUE(solver.dt > 0.);
    ASSERT_TRUE(solver.npar == 135 && solver.N == 20);
    for (int k = 0; k < solver.N; k++) solver.setParameter(k, "reference_velocity", 1.);
    for (int k = 0; k < solver.N; k++) ASSERT_TRUE(solver.getParameter(k, "reference_velocity") == 1.);
    ASSERT_TRUE(solver.hasParameter("lin_constraint_7_b") && !solver.hasParameter("nope"));
    solver.setXinit("x", 5.4); solver.setXinit("y", 1.4);
    double sum_init = 0.;
    for (unsigned int i = 0; i < solver.nx; i++) sum_init += solver._params.xinit[i];
    ASSERT_TRUE(std::abs(sum_init - 6.8) < 1e-5);
    State state;
    state.set("x", 4.4); state.set("y", 1.2);
    solver.setXinit(state);
    sum_init = 0.;
    for (unsigned int i = 0; i < solver.nx; i++) sum_init += solver._params.xinit[i];
    ASSERT_TRUE(std::abs(sum_init - 5.6) < 1e-5);
    state = State(); solver.setXinit(state);
    for (int k = 0; k < solver.N; k++) { solver.setEgoPrediction(k, "x", k * 1.); solver.setEgoPrediction(k, "y", 0.); }
    for (int k = 0; k < solver.N; k++) { ASSERT_TRUE(solver.getEgoPrediction(k, "x") == k * 1); ASSERT_TRUE(solver.getEgoPrediction(k, "y") == 0); }
    ASSERT_TRUE(solver._params.x0[3 * solver.nvar + 2] == 3.0);           // layout [u_k; x_k], x is z-index 2
    // warm start from the previous output (acados flavour of test_solver.cpp:119-127)
    for (int k = 0; k <= solver.N; k++) solver._output.xtraj[k * solver.nx + 0] = 10. + k;
    solver.initializeWarmstart(state, false);
    ASSERT_TRUE(solver.getEgoPrediction(1, "x") == 11.0 && solver.getEgoPrediction(0, "x") == 10.0);
    state.set("x", 0.8);
    solver.initializeWarmstart(state, true);
    ASSERT_TRUE(solver.getEgoPrediction(0, "x") == 0.8 && solver.getEgoPrediction(1, "x") == 12.0);
    ASSERT_TRUE(solver.getEgoPrediction(solver.N, "x") == 10. + solver.N - 1);
    // node 0 of the shifted warm start: states from `state`, inputs 0 (the reference reads State::get(<input>) out of bounds)
    solver.setEgoPrediction(0, "a", 123.); solver.setEgoPrediction(0, "w", -4.);
    solver.initializeWarmstart(state, true);
    ASSERT_TRUE(solver.getEgoPrediction(0, "a") == 0.0 && solver.getEgoPrediction(0, "w") == 0.0);
    ASSERT_TRUE(state.get("a") == 0.0);
    state.set("w", 5.0);                                                  // not a state: ignored, nothing overwritten
    ASSERT_TRUE(state.get("x") == 0.8 && state.get("w") == 0.0);
    // generated fast setters hit the same slots as the string interface
    setSolverParameterEllipsoidObstX(3, solver._params, 7.25, 2);
    ASSERT_TRUE(solver.getParameter(3, "ellipsoid_obst_2_x") == 7.25);
    setSolverParameterLinConstraintB(4, solver._params, -1.5, 7);
    ASSERT_TRUE(solver.getParameter(4, "lin_constraint_7_b") == -1.5);
    Solver solver2(1);
    solver2.setParameter(0, "reference_velocity", 3.);
    ASSERT_TRUE(solver._solver_id != solver2._solver_id);
    solver2 = solver;
    ASSERT_TRUE(solver2.getParameter(0, "reference_velocity") == 1.);
    ASSERT_TRUE(solver.explainExitFlag(1) == "Success");
    // deterministic stand-in for the wall-clock exit of Solver::solve (:111-116)
    ASSERT_TRUE(solver.iterationBudget() == solver._num_iterations);
    solver.setIterationTimeEstimate(0.004); solver._params.solver_timeout = 0.02;
    ASSERT_TRUE(solver.iterationBudget() == std::min(5, solver._num_iterations));
    solver._params.solver_timeout = 0.001;
    ASSERT_TRUE(solver.iterationBudget() == 1);
    solver.setIterationTimeEstimate(0.);
    std::printf("plumbing ok\n");
    if (!do_solve) return 0;

    // ---- GPU: a straight-road problem with far-away obstacles must succeed; solve() == solveBatch() ----
    std::vector<std::unique_ptr<Solver>> own; std::vector<Solver *> batch;
    for (int b = 0; b < 3; b++) {
        own.emplace_back(new Solver(b)); Solver &s = *own.back();
        State st; st.set("v", 1.0 + 0.2 * b); s.setXinit(st);
        s.initializeWithState(st);
        for (int k = 0; k <= s.N; k++) { s.setEgoPrediction(k, "x", 0.2 * k * (1.0 + 0.2 * b)); s.setEgoPrediction(k, "spline", 0.2 * k * (1.0 + 0.2 * b)); }
        const char *w[] = {"acceleration", "angular_velocity", "velocity", "reference_velocity", "contour", "lag"};
        const double wv[] = {0.34, 0.85, 0.55, 2.0, 0.05, 0.75};
        for (int k = 0; k < s.N; k++) {
            for (int i = 0; i < 6; i++) s.setParameter(k, std::string(w[i]), wv[i]);
            for (int i = 0; i < SOLVER_S; i++) { setSolverParameterSplineXC(k, s._params, 1.0, i); setSolverParameterSplineXD(k, s._params, 6.0 * i, i); setSolverParameterSplineStart(k, s._params, 6.0 * i, i); }
            setSolverParameterEgoDiscRadius(k, s._params, 0.325);
            for (int j = 0; j < SOLVER_MAX_OBSTACLES; j++) {     // (SOLVER_M is 0 in generated libraries: rows are not typed there)
                setSolverParameterLinConstraintA1(k, s._params, 1.0, j); setSolverParameterLinConstraintB(k, s._params, 100.0, j);
                setSolverParameterEllipsoidObstX(k, s._params, 50.0, j); setSolverParameterEllipsoidObstY(k, s._params, 50.0 + j, j);
                setSolverParameterEllipsoidObstChi(k, s._params, 1.0, j); setSolverParameterEllipsoidObstR(k, s._params, 0.4, j);
            }
        }
        batch.push_back(&s);
    }
    BatchContext ctx;
    std::vector<int> codes = Solver::solveBatch(ctx, batch);
    for (int b = 0; b < 3; b++) ASSERT_TRUE(ctx.slotOf(batch[b]) == b && batch[b]->_info.solvetime > 0. && batch[b]->_info.min_time < 1e11);   // (:151-153) timing fields filled
    for (int b = 0; b < 3; b++) {
        ASSERT_TRUE(codes[b] == 1);
        Solver single(10 + b); single = *batch[b];
        single.loadWarmstart();
        ASSERT_TRUE(single.solve() == 1);
        ASSERT_TRUE(single._info.pobj == batch[b]->_info.pobj);          // bitwise: same kernel, batch-composition invariant
        for (int k = 0; k <= single.N; k++) ASSERT_TRUE(single.getOutput(k, "x") == batch[b]->getOutput(k, "x"));
        ASSERT_TRUE(single.getOutput(5, "v") > 1.0 && single.getOutput(single.N, "x") > 3.0);
        // Forces-style indexing of the same solution (forces_solver_interface.cpp:241-244): entry i of z_k = [u_k; x_k], k < N
        ASSERT_TRUE(single.getForcesStyleOutput(3, 0) == single.getOutput(3, "a") && single.getForcesStyleOutput(3, 1) == single.getOutput(3, "w"));
        ASSERT_TRUE(single.getForcesStyleOutput(7, 2) == single.getOutput(7, "x") && single.getForcesStyleOutput(7, 5) == single.getOutput(7, "v"));
        // ... and as a MODE: N stages, Forces warm-start loops (forces_solver_interface.cpp:147-182), this solver's node N repeats the terminal stage
        ASSERT_TRUE(single.stages() == single.N + 1);
        single.setStageIndexing(Solver::StageIndexing::ForcesStages);
        ASSERT_TRUE(single.stages() == single.N && single.getOutput(single.N - 1, "x") == single.getForcesStyleOutput(single.N - 1, 2));
        State now; now.set("x", 0.25); now.set("v", 1.5);
        single.initializeWarmstart(now, true);
        ASSERT_TRUE(single.getEgoPrediction(0, "x") == 0.25 && single.getEgoPrediction(3, "y") == single.getOutput(4, "y"));
        ASSERT_TRUE(single.getEgoPrediction(single.N - 1, "x") == single.getOutput(single.N - 1, "x") && single.getEgoPrediction(single.N, "x") == single.getOutput(single.N - 1, "x"));
        single.initializeWarmstart(now, false);
        ASSERT_TRUE(single.getEgoPrediction(3, "y") == single.getOutput(3, "y") && single.getEgoPrediction(single.N, "v") == single.getOutput(single.N - 1, "v"));
        single.setStageIndexing(Solver::StageIndexing::AcadosNodes);
    }
    // ---- the one-iteration protocol (initializeOneIteration / solveOneIteration x n / completeOneIteration, :121-204; what
    // SH-MPC's scenario module drives, scenario_constraints.cpp:85) gives bitwise what solve() gives ----
    {
        Solver a(20), b(21);
        a = *batch[1]; b = *batch[1];
        a.loadWarmstart(); b.loadWarmstart();
        const int ea = a.solve();
        b.initializeOneIteration();
        int n_done = 0;
        for (int it = 0; it < b._num_iterations; it++) {
            const int st = b.solveOneIteration();
            n_done++;
            ASSERT_TRUE(st == 0);
            if (b._info.qp_status != 0) break;
        }
        const int eb = b.completeOneIteration();
        ASSERT_TRUE(ea == 1 && eb == 1 && n_done == b._num_iterations && b._info.sqp_iter == a._info.sqp_iter);
        ASSERT_TRUE(a._info.pobj == b._info.pobj);
        for (int k = 0; k <= a.N; k++) { ASSERT_TRUE(a.getOutput(k, "x") == b.getOutput(k, "x")); ASSERT_TRUE(a.getOutput(k, "v") == b.getOutput(k, "v")); }
        // a second solve() of the same solver without loadWarmstart() continues from its own iterate and multipliers (the capsule's
        // state persists): more RTI iterations on the same problem move the objective towards the converged optimum, not away
        const double first = a._info.pobj;
        ASSERT_TRUE(a.solve() == 1);
        ASSERT_TRUE(a._info.pobj <= first + 1e-9 && std::fabs(a._info.pobj - first) < 1e-2);
        // with loadWarmstart() the primal iterate comes from _params.x0 again, the multipliers stay: not the fresh result bitwise
        a.loadWarmstart();
        ASSERT_TRUE(a.solve() == 1);
        ASSERT_TRUE(std::fabs(a._info.pobj - first) < 1e-3);
    }
    // ---- per-caller batch contexts (round-2 verdict): a Solver keeps ITS state slot whatever subset of a module's solvers a tick
    // launches and whatever another module instance does in between; the reference has one capsule per Solver (:17,51-65) ----
    {
        auto fresh = [&](int id, int like) { std::unique_ptr<Solver> s(new Solver(id)); *s = *batch[like]; return s; };
        // reference run: solver P alone in its own context, two ticks (multipliers carried from tick 1 to tick 2)
        auto p_ref = fresh(30, 1);
        BatchContext c_ref;
        p_ref->loadWarmstart(); ASSERT_TRUE(Solver::solveBatch(c_ref, {p_ref.get()})[0] == 1);
        p_ref->loadWarmstart(); ASSERT_TRUE(Solver::solveBatch(c_ref, {p_ref.get()})[0] == 1);
        const double want1 = p_ref->_info.pobj;
        // the same solver P as entry 2 of a three-solver tick, then ALONE (entry 0) in the next tick of the same context, while a second
        // context solves other problems in between: P's second tick must see exactly P's multipliers from its first tick
        auto q0 = fresh(31, 0), q2 = fresh(32, 2), p = fresh(33, 1), other = fresh(34, 2);
        BatchContext c1, c2;
        for (Solver *s : {q0.get(), q2.get(), p.get(), other.get()}) s->loadWarmstart();
        std::vector<int> t1 = Solver::solveBatch(c1, {q0.get(), q2.get(), p.get()});
        ASSERT_TRUE(t1[2] == 1 && c1.slotOf(p.get()) == 2);
        ASSERT_TRUE(Solver::solveBatch(c2, {other.get()})[0] == 1);                 // another module instance, its own handle and slots
        p->loadWarmstart();
        std::vector<int> t2 = Solver::solveBatch(c1, {p.get()});                     // a tick that launches only P: batch entry 0, still slot 2
        ASSERT_TRUE(t2[0] == 1 && c1.slotOf(p.get()) == 2);
        ASSERT_TRUE(p->_info.pobj == want1);                                         // bitwise: its own multipliers, nobody else's
        // (how sensitive this is depends on the problem: with every constraint inactive the multipliers of the rows are zero and only
        // the dynamics multipliers distinguish the slots; the device-side slot map itself is tested in tests/test_gpu_iterations.py)
        // growth: more solvers than the context's first capacity keep every earlier slot's state
        std::vector<std::unique_ptr<Solver>> many; std::vector<Solver *> mb{p.get()};
        for (int i = 0; i < 20; i++) { many.push_back(fresh(40 + i, i % 3)); many.back()->loadWarmstart(); mb.push_back(many.back().get()); }
        const int cap0 = c1.capacity();
        p->loadWarmstart();
        auto p_ref3 = p_ref.get(); p_ref3->loadWarmstart(); Solver::solveBatch(c_ref, {p_ref3});      // reference: third tick of P alone
        std::vector<int> t3 = Solver::solveBatch(c1, mb);
        ASSERT_TRUE(c1.capacity() > cap0 && c1.slotOf(p.get()) == 2 && t3[0] == 1);
        ASSERT_TRUE(p->_info.pobj == p_ref3->_info.pobj);
        std::printf("batch contexts ok: capacity %d -> %d\n", cap0, c1.capacity());
    }
    // ---- slot lifetime (round-3 advisor): a destroyed Solver releases its slot, the slot is CLEARED before it is reused (a new Solver
    // -- also one that lands at the same address -- starts like a new capsule), and a Solver named twice is solved once ----
    {
        auto fresh = [&](int id, int like) { std::unique_ptr<Solver> s(new Solver(id)); *s = *batch[like]; return s; };
        BatchContext c;
        auto keep = fresh(60, 0);
        keep->loadWarmstart();
        // first tick of a FRESH solver on problem 1, alone in its own context: the reference value for "no inherited multipliers"
        double first_tick;
        { auto f = fresh(61, 1); BatchContext cf; f->loadWarmstart(); ASSERT_TRUE(Solver::solveBatch(cf, {f.get()})[0] == 1); first_tick = f->_info.pobj; }
        int slot_a;
        {
            auto a = fresh(62, 1);
            a->loadWarmstart();
            ASSERT_TRUE(Solver::solveBatch(c, {keep.get(), a.get()})[1] == 1);
            a->loadWarmstart();
            ASSERT_TRUE(Solver::solveBatch(c, {keep.get(), a.get()})[1] == 1);        // second tick: a's slot now holds multipliers
            slot_a = c.slotOf(a.get());
            ASSERT_TRUE(slot_a == 1);
        }                                                                           // ~Solver: the slot goes back to the context's pool
        auto b = fresh(63, 1);
        b->loadWarmstart(); keep->loadWarmstart();
        std::vector<int> t = Solver::solveBatch(c, {keep.get(), b.get(), b.get()});    // (b twice: solved once, both mentions get its code)
        ASSERT_TRUE(t.size() == 3 && t[1] == 1 && t[2] == t[1]);
        ASSERT_TRUE(c.slotOf(b.get()) == slot_a);                                    // the freed slot, reused ...
        ASSERT_TRUE(b->_info.pobj == first_tick);                                   // ... and cleared: bitwise a new capsule's first tick
        std::printf("slot reuse ok: slot %d\n", slot_a);
    }
    std::printf("solve ok: pobj %.6f %.6f %.6f\n", batch[0]->_info.pobj, batch[1]->_info.pobj, batch[2]->_info.pobj);
    return 0;
}
