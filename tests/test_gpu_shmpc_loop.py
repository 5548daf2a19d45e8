"""GPU: an SH-MPC closed loop that stays on the device between ticks (SURVEY 8 f-2 + f-3 + the solve + the selection):
warm start of every parallel scenario solver from the chosen plan (tmpc_warmstart), polygon of each solver's own sampled scenarios
around that warm start (tmpc_scenario_halfspaces), ten RTI iterations with the multipliers the solver kept (tmpc_solve_iterations),
support of the solutions (tmpc_scenario_support), lowest-objective selection (scenario_constraints.cpp:38-108).  Every tick is
compared with the host mirrors (bitwise: warm start, rows) and with the oracle driven the same way (solutions, exit codes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shmpc_closed_loop_on_device():
    import torch
    import oracle_lib as O
    from mpc_planner_amd import scenes, solver, modules as md
    N, P, R, TICKS = 20, 8, 24, 5
    radius = scenes.OBSTACLE_RADIUS + scenes.ROBOT_RADIUS
    sc = scenes.make_scene(41, N=N, M=8, B=P, slack=True, n_scenario=R)
    pm = sc["pm"]
    M, S_cen = sc["samples"].shape[:2]
    rng = np.random.default_rng(5)
    # every parallel solver draws its own scenarios (scenario_constraints.cpp:121-131): the scene's samples plus a per-solver draw
    base = sc["samples"][None] + rng.normal(0.0, 0.05, (P,) + sc["samples"].shape)          # [P][M][S_cen][N][2]
    dims = solver.default_dims(N=N, S=5, n_lin=0, M=0, n_slk=R, slack=1)
    s = solver.BatchedSolver(dims, B_max=P)
    pb = O.problem(N=N, S=5, n_lin=0, M=0, n_slk=R, slack=1)
    dev = torch.device("cuda")
    # the main solver's warm start and parameters, copied to every parallel solver (:73)
    xinit = np.repeat(sc["xinit"][:1], P, 0); x0 = np.repeat(sc["x0"][:1], P, 0); params = np.repeat(sc["params"][:1], P, 0)
    for j in range(R):                                             # the scenario rows come from the device
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_a1")] = 1.0
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_a2")] = 0.0
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_b")] = 1e3
    s.set_batch(xinit, x0, params)
    t_scene_of = torch.arange(P, dtype=torch.int32, device=dev)
    pi = np.zeros((P, (N + 1) * 5)); lamh = np.zeros((P, N * O.MAX_NH))
    n_success = 0
    for tick in range(TICKS):
        # the obstacles moved on: prediction step k of this tick is step k + tick of the first one
        kk = np.minimum(np.arange(N) + tick, N - 1)
        smp = base[:, :, :, kk, :]                                                           # [P][M][S_cen][N][2]
        dsm = np.ascontiguousarray(smp.transpose(0, 3, 1, 2, 4)).reshape(P, N, M * S_cen, 2)
        t_s = torch.from_numpy(dsm).to(dev)
        t_sx = torch.from_numpy(np.ascontiguousarray(xinit[:, 0])).to(dev)
        s.scenario_halfspaces(t_s.data_ptr(), M * S_cen, R, t_scene_of.data_ptr(), t_sx.data_ptr(), radius)
        # host mirror of the rows, bitwise
        want = params.copy(); which = []
        for p_ in range(P):
            rows = md.scenario_halfspaces(x0[p_], smp[p_], radius, R, return_index=True)
            md.halfspace_rows_set_parameters(pm, want[p_], xinit[p_, 0], rows[:3], "disc_0_scenario_constraint", R)
            which.append(rows[3])
        got = s.debug_get_params()
        assert np.array_equal(got, want), tick
        s.solve_iterations(10, keep_iterate=False, keep_multipliers=True)
        g = s.get()
        sup, act = s.scenario_support(S_cen, 1e-3)
        # the oracle, driven the same way (its own multipliers, reset on failure like the device's)
        ec = np.zeros(P, np.int32); xt = np.zeros((P, N + 1, 6))
        for p_ in range(P):
            xt[p_], _, info = O.solve_carry(pb, xinit[p_], x0[p_], want[p_], 10, pi[p_], lamh[p_])
            ec[p_] = info.exit_code
        assert (g["exit_code"] == ec).all(), (tick, g["exit_code"], ec)
        ok = ec == 1
        n_success += int(ok.sum())
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        assert (np.abs(g["xtraj"][ok] - xt[ok]) / sx).max() < 1e-6, tick
        for p_ in np.flatnonzero(ok):                              # support: the device's count on the device's solution == the mirror's
            assert (int(sup[p_]), int(act[p_])) == md.scenario_support(g["xtraj"][p_], want[p_], pm, which[p_], S_cen, 1e-3), (tick, p_)
        assert ok.any(), tick
        best = int(np.flatnonzero(ok)[np.argmin(g["pobj"][ok])])   # :93-107
        assert best == s.select_best()
        # next tick: the robot moved one step along the chosen plan; every solver restarts from that plan, shifted (:73, :82)
        state = g["xtraj"][best, 1].copy()
        t_state = torch.from_numpy(np.tile(state, (P, 1))).to(dev)
        t_mode = torch.ones(P, dtype=torch.int32, device=dev); t_src = torch.full((P,), best, dtype=torch.int32, device=dev)
        s.warmstart(t_state.data_ptr(), t_mode.data_ptr(), t_src.data_ptr())
        x0_dev, xinit_dev = s.debug_get_x0()
        for p_ in range(P):
            x0[p_] = md.initialize_warmstart(x0[p_].copy(), state, g["xtraj"][best], g["utraj"][best], shift_previous_solution_forward=True)
            xinit[p_] = state
        assert np.array_equal(x0_dev, x0) and np.array_equal(xinit_dev, xinit), tick
    assert n_success >= TICKS * P // 2
    s.close()
