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


def _mixture_prediction(obs_pos, n_extra, dt=0.2):
    from mpc_planner_amd import scenes
    return scenes.mixture_prediction(obs_pos, n_extra, dt)


def test_device_sampler_and_removal_match_host_mirrors():
    """f-3: the scenario sampler (counter-based, bit-reproducible) and the scenario-removal step on device against their host mirrors."""
    import torch
    from mpc_planner_amd import scenes, solver, modules as md
    N, P, R, S_cen, M = 20, 6, 24, 256, 8
    radius = scenes.OBSTACLE_RADIUS + scenes.ROBOT_RADIUS
    sc = scenes.make_scene(41, N=N, M=M, B=P, slack=True, n_scenario=R)
    pm = sc["pm"]
    dev = torch.device("cuda")
    pred = np.repeat(_mixture_prediction(sc["obstacles"]["pos"], 0)[None], P, 0)             # [P][M][3][N][6]: every solver sees the same prediction
    prob = np.tile([0.5, 0.25, 0.25], (P, M, 1))
    s = solver.BatchedSolver(solver.default_dims(N=N, S=5, n_lin=0, M=0, n_slk=R, slack=1), B_max=P)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    t_pred = torch.from_numpy(pred).to(dev); t_prob = torch.from_numpy(prob).to(dev)
    t_smp = torch.empty((P, N, M * S_cen, 2), dtype=torch.float64, device=dev)
    s.sample_scenarios(t_pred.data_ptr(), t_prob.data_ptr(), P, M, 3, S_cen, 777, t_smp.data_ptr())
    s.synchronize()
    got = t_smp.cpu().numpy()
    want = md.sample_scenarios(pred, prob, S_cen, 777)
    assert np.array_equal(got, want)                                     # bit for bit
    assert not np.array_equal(got[0], got[1])                            # every solver draws its own scenarios (:121-131)
    # the draws have the moments they should: per obstacle and step, mean of the mixture and spread of the order of the radii
    smp = got.reshape(P, N, M, S_cen, 2)
    mix_mean = (pred[..., 0:2] * prob[..., None, None]).sum(2)          # [P][M][N][2]
    err = np.abs(smp.mean(3).transpose(0, 2, 1, 3) - mix_mean).max()
    assert err < 0.15, err
    # removal: device marks == mirror's; the polygon rows built without the discarded scenarios == mirror's, bit for bit
    t_scene_of = torch.arange(P, dtype=torch.int32, device=dev); t_sx = torch.from_numpy(np.ascontiguousarray(sc["xinit"][:, 0])).to(dev)
    n_discard = 7
    s.scenario_discard(t_smp.data_ptr(), M * S_cen, S_cen, n_discard, t_scene_of.data_ptr(), radius)
    marks = s.scenario_discarded(S_cen)
    rows_want = sc["params"].copy()
    for p_ in range(P):
        smp_p = want[p_].reshape(N, M, S_cen, 2).transpose(1, 2, 0, 3)                      # [M][S_cen][N][2]
        mk = md.scenario_discard(sc["x0"][p_], smp_p, radius, n_discard)
        assert np.array_equal(marks[p_], mk) and mk.sum() == n_discard
        rows = md.scenario_halfspaces(sc["x0"][p_], smp_p, radius, R, return_index=True, discard=mk)
        md.halfspace_rows_set_parameters(pm, rows_want[p_], sc["xinit"][p_, 0], rows[:3], "disc_0_scenario_constraint", R)
        assert not np.isin(rows[3][rows[3] >= 0] % S_cen, np.flatnonzero(mk)).any()       # no row comes from a discarded scenario
    s.scenario_halfspaces(t_smp.data_ptr(), M * S_cen, R, t_scene_of.data_ptr(), t_sx.data_ptr(), radius)
    assert np.array_equal(s.debug_get_params(), rows_want)
    # a new batch forgets the marks: the next polygons use every scenario again
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.scenario_halfspaces(t_smp.data_ptr(), M * S_cen, R, t_scene_of.data_ptr(), t_sx.data_ptr(), radius)
    full = sc["params"].copy()
    for p_ in range(P):
        smp_p = want[p_].reshape(N, M, S_cen, 2).transpose(1, 2, 0, 3)
        md.halfspace_rows_set_parameters(pm, full[p_], sc["xinit"][p_, 0], md.scenario_halfspaces(sc["x0"][p_], smp_p, radius, R), "disc_0_scenario_constraint", R)
    assert np.array_equal(s.debug_get_params(), full)
    s.close()


def test_shmpc_closed_loop_with_device_sampled_scenarios():
    """The SH-MPC loop with NO host-generated samples (round-2 verdict, f-3): every tick the scenarios of every parallel solver are
    drawn on the device from the obstacles' mixture predictions (a few KB per solver instead of 84 MB of samples per 4096-trajectory
    launch), the most constraining ones are removed, polygons, ten RTI iterations, support with the removed scenarios counted into
    the risk bound, selection, warm start -- all on the device; the host only mirrors it for the comparison."""
    import torch
    import oracle_lib as O
    from mpc_planner_amd import scenes, solver, modules as md
    N, P, R, TICKS, S_cen, M, N_DISCARD = 20, 8, 24, 4, 256, 8, 4
    radius = scenes.OBSTACLE_RADIUS + scenes.ROBOT_RADIUS
    sc = scenes.make_scene(41, N=N, M=M, B=P, slack=True, n_scenario=R)
    pm = sc["pm"]
    long_pred = _mixture_prediction(sc["obstacles"]["pos"], TICKS)                           # [M][3][N + TICKS][6]
    prob = np.tile([0.5, 0.25, 0.25], (P, M, 1))
    dims = solver.default_dims(N=N, S=5, n_lin=0, M=0, n_slk=R, slack=1)
    s = solver.BatchedSolver(dims, B_max=P)
    pb = O.problem(N=N, S=5, n_lin=0, M=0, n_slk=R, slack=1)
    dev = torch.device("cuda")
    xinit = np.repeat(sc["xinit"][:1], P, 0); x0 = np.repeat(sc["x0"][:1], P, 0); params = np.repeat(sc["params"][:1], P, 0)
    for j in range(R):
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_a1")] = 1.0
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_a2")] = 0.0
        params[:, :, pm.index(f"disc_0_scenario_constraint_{j}_b")] = 1e3
    s.set_batch(xinit, x0, params)
    t_scene_of = torch.arange(P, dtype=torch.int32, device=dev)
    t_prob = torch.from_numpy(prob).to(dev)
    t_smp = torch.empty((P, N, M * S_cen, 2), dtype=torch.float64, device=dev)
    pi = np.zeros((P, (N + 1) * 5)); lamh = np.zeros((P, N * O.MAX_NH))
    n_success = 0
    for tick in range(TICKS):
        pred = np.repeat(long_pred[None, :, :, tick:tick + N, :], P, 0)                      # the obstacles moved on one step
        t_pred = torch.from_numpy(np.ascontiguousarray(pred)).to(dev)                        # 8 x 8 x 3 x 20 x 6 doubles: the PREDICTION, not samples
        s.sample_scenarios(t_pred.data_ptr(), t_prob.data_ptr(), P, M, 3, S_cen, 1000 + tick, t_smp.data_ptr())
        t_sx = torch.from_numpy(np.ascontiguousarray(xinit[:, 0])).to(dev)
        s.scenario_discard(t_smp.data_ptr(), M * S_cen, S_cen, N_DISCARD, t_scene_of.data_ptr(), radius)
        s.scenario_halfspaces(t_smp.data_ptr(), M * S_cen, R, t_scene_of.data_ptr(), t_sx.data_ptr(), radius)
        # host mirror of sampler -> removal -> rows, bitwise
        smp_all = md.sample_scenarios(pred, prob, S_cen, 1000 + tick)
        want = params.copy(); which = []
        for p_ in range(P):
            smp_p = smp_all[p_].reshape(N, M, S_cen, 2).transpose(1, 2, 0, 3)
            mk = md.scenario_discard(x0[p_], smp_p, radius, N_DISCARD)
            rows = md.scenario_halfspaces(x0[p_], smp_p, radius, R, return_index=True, discard=mk)
            md.halfspace_rows_set_parameters(pm, want[p_], xinit[p_, 0], rows[:3], "disc_0_scenario_constraint", R)
            which.append(rows[3])
        assert np.array_equal(s.debug_get_params(), want), tick
        s.solve_iterations(10, keep_iterate=False, keep_multipliers=True, new_solve=True)
        g = s.get()
        sup, act = s.scenario_support(S_cen, 1e-3)
        ec = np.zeros(P, np.int32); xt = np.zeros((P, N + 1, 6))
        for p_ in range(P):
            xt[p_], _, info = O.solve_carry(pb, xinit[p_], x0[p_], want[p_], 10, pi[p_], lamh[p_])
            ec[p_] = info.exit_code
        assert (g["exit_code"] == ec).all(), (tick, g["exit_code"], ec)
        ok = ec == 1
        n_success += int(ok.sum())
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        assert (np.abs(g["xtraj"][ok] - xt[ok]) / sx).max() < 1e-6, tick
        for p_ in np.flatnonzero(ok):
            assert (int(sup[p_]), int(act[p_])) == md.scenario_support(g["xtraj"][p_], want[p_], pm, which[p_], S_cen, 1e-3), (tick, p_)
            # the certificate: removed scenarios count into the compression set
            assert md.scenario_risk(S_cen, int(sup[p_]), removed=N_DISCARD) >= md.scenario_risk(S_cen, int(sup[p_]))
        assert ok.any(), tick
        best = int(np.flatnonzero(ok)[np.argmin(g["pobj"][ok])])
        assert best == s.select_best()
        state = g["xtraj"][best, 1].copy()
        t_state = torch.from_numpy(np.tile(state, (P, 1))).to(dev)
        t_mode = torch.ones(P, dtype=torch.int32, device=dev); t_src = torch.full((P,), best, dtype=torch.int32, device=dev)
        s.warmstart(t_state.data_ptr(), t_mode.data_ptr(), t_src.data_ptr())
        for p_ in range(P):
            x0[p_] = md.initialize_warmstart(x0[p_].copy(), state, g["xtraj"][best], g["utraj"][best], shift_previous_solution_forward=True)
            xinit[p_] = state
    assert n_success >= TICKS * P // 2
    s.close()
