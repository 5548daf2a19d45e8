"""GPU: the reference's "one iteration at a time" protocol and multipliers carried across control ticks
(tmpc_solve_iterations / tmpc_reset_multipliers; acados_solver_interface.cpp:67-77,121-204,274-284)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solver(mode, B_max, **pkw):
    from mpc_planner_amd import solver
    s = solver.BatchedSolver(solver.default_dims(**pkw), B_max=B_max)
    if mode == "latency":
        s.set_latency_mode(True)
    elif mode == "latency2":                                           # parallel-in-time Newton solve (csrc/tmpc_scan.hpp)
        assert s.set_latency_mode(2)
    elif mode == "lanes":
        s.set_throughput_mode(True)
    return s


@pytest.mark.parametrize("mode", ["wave", "latency", "latency2", pytest.param("lanes", marks=pytest.mark.lanes)])
def test_one_iteration_calls_equal_one_solve(mode):
    """10 x solveOneIteration == solve(): bitwise, including trajectories whose loop ends early (QP at its iteration limit)
    and infeasible ones."""
    from mpc_planner_amd import scenes
    import test_gpu_parity as T
    sc = T._make_infeasible(scenes.make_scene(7, N=20, M=8, B=64), [9, 33])
    s = _solver(mode, 64, N=20, S=5, n_lin=8, M=8, qp_iter_max=6)          # a low QP iteration limit: some loops end early
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); ref = s.get()
    assert (ref["sqp_iter"] < 10).any() and (ref["sqp_iter"] == 10).any() and (ref["exit_code"] != 1).any()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve_iterations(10); one = s.get()
    for k in ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "sqp_iter", "qp_iter_total", "res_eq"):
        np.testing.assert_array_equal(one[k], ref[k], err_msg=k)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.solve_iterations(1, complete=False)                          # initializeOneIteration + solveOneIteration
    for i in range(9):
        s.solve_iterations(1, keep_iterate=True, keep_multipliers=True, complete=(i == 8))
    g = s.get()                                                        # (sqp_iter / qp_iter_total count the last call only)
    for k in ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "res_eq"):
        np.testing.assert_array_equal(g[k], ref[k], err_msg=k)
    s.close()


@pytest.mark.parametrize("mode", ["wave", pytest.param("lanes", marks=pytest.mark.lanes)])
def test_multipliers_carried_across_ticks_match_oracle(mode):
    """Closed loop of 6 control ticks: every tick loads a shifted warm start (loadWarmstart: primal only) and keeps the
    multipliers of the slot's previous solve, like the reference's capsules; a trajectory made infeasible at tick 2 gets its
    multipliers reset (Solver_acados_reset).  The oracle carries (and resets) its multipliers the same way."""
    import oracle_lib as O
    from mpc_planner_amd import scenes, modules as md
    B, N = 16, 20
    sc = scenes.make_scene(12, N=N, M=8, B=B)
    pm = sc["pm"]
    pb = O.problem(N=N, S=5, n_lin=8, M=8)
    s = _solver(mode, B, N=N, S=5, n_lin=8, M=8)
    xinit, x0, params = sc["xinit"].copy(), sc["x0"].copy(), sc["params"].copy()
    pi = np.zeros((B, (N + 1) * 5)); lamh = np.zeros((B, N * O.MAX_NH))
    fresh = O.problem(N=N, S=5, n_lin=8, M=8)
    differs_from_fresh = 0.0
    for tick in range(6):
        p_t = params.copy()
        if tick == 2:                                              # contradictory topology rows on one trajectory for this tick only
            for j, sg in ((0, 1.0), (1, -1.0)):
                p_t[5, 1:, pm.index(f"lin_constraint_{j}_a1")] = sg
                p_t[5, 1:, pm.index(f"lin_constraint_{j}_a2")] = 0.0
                p_t[5, 1:, pm.index(f"lin_constraint_{j}_b")] = sg * x0[5, 1:-1, 2] - 5.0
        s.set_batch(xinit, x0, p_t)
        s.solve_iterations(10, keep_iterate=False, keep_multipliers=True)
        g = s.get()
        xt = np.zeros((B, N + 1, 5)); ut = np.zeros((B, N, 2)); ec = np.zeros(B, np.int32)
        for b in range(B):
            xt[b], ut[b], info = O.solve_carry(pb, xinit[b], x0[b], p_t[b], 10, pi[b], lamh[b])
            ec[b] = info.exit_code
            if tick >= 1 and info.exit_code == 1:
                xf, uf, inf_f = O.solve(fresh, xinit[b], x0[b], p_t[b])
                differs_from_fresh = max(differs_from_fresh, np.abs(xf - xt[b]).max())
        assert (g["exit_code"] == ec).all(), (tick, g["exit_code"], ec)
        if tick == 2:
            assert ec[5] != 1 and not pi[5].any() and not lamh[5].any()
        ok = ec == 1
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        assert (np.abs(g["xtraj"][ok] - xt[ok]) / sx).max() < 1e-6, tick
        # next tick: the robot moves to the first predicted state of its own plan; warm start = shifted previous solution
        for b in range(B):
            src_x, src_u = (xt[b], ut[b]) if ok[b] else (x0[b][:, 2:], x0[b][:-1, :2])
            state = src_x[1].copy()
            xinit[b] = state
            x0[b] = md.initialize_warmstart(x0[b].copy(), state, src_x, src_u, shift_previous_solution_forward=True)
    assert differs_from_fresh > 1e-7          # carrying the multipliers matters (exact Hessian of the first RTI iteration)
    s.close()


def test_reset_multipliers_and_other_iteration_budget():
    """tmpc_reset_multipliers == a new capsule; n_iter = 4 (mpc_planner_rosnavigation's iterations) against the oracle."""
    import oracle_lib as O
    from mpc_planner_amd import scenes
    sc = scenes.make_scene(3, N=20, M=8, B=32)
    s = _solver("wave", 32, N=20, S=5, n_lin=8, M=8)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve_iterations(4); a = s.get()
    pb = O.problem(N=20, S=5, n_lin=8, M=8, n_sqp=4)
    xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(32, -1), sc["params"].reshape(32, -1))
    assert (a["exit_code"] == info["exit_code"]).all() and (a["sqp_iter"] == info["sqp_iter"]).all()
    ok = info["exit_code"] == 1
    np.testing.assert_allclose(a["xtraj"][ok], xt[ok], rtol=0, atol=1e-7)
    s.solve_iterations(4, keep_multipliers=True); b = s.get()          # same warm start, kept multipliers: a different iterate
    assert np.abs(b["xtraj"] - a["xtraj"]).max() > 1e-9
    s.reset_multipliers()
    s.solve_iterations(4, keep_multipliers=True); c = s.get()          # after the reset: the fresh result again, bitwise
    np.testing.assert_array_equal(c["xtraj"], a["xtraj"])
    s.close()


@pytest.mark.parametrize("mode", ["wave", pytest.param("lanes", marks=pytest.mark.lanes)])
def test_new_solve_reopens_loops_that_ended(mode):
    """Advisor (round 2): a slot whose last QP stopped with qp_status != 0 must iterate again in the NEXT solve() even without a
    loadWarmstart -- the reference's loop exit (:105-106) is local to one solve().  Without TMPC_ITER_NEW_SOLVE the slot would be
    skipped and the previous trajectory returned as a fresh result."""
    from mpc_planner_amd import scenes
    sc = scenes.make_scene(7, N=20, M=8, B=64)
    s = _solver(mode, 64, N=20, S=5, n_lin=8, M=8, qp_iter_max=6)          # a low QP iteration limit: some loops end early (qp_status 2)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve_iterations(10); a = s.get()
    early = a["qp_status"] != 0
    assert early.any() and (~early).any()
    # second solve() of the same Solvers, iterate kept (no loadWarmstart): every slot must run again
    s.solve_iterations(10, keep_iterate=True, keep_multipliers=True, new_solve=True); b = s.get()
    assert (b["sqp_iter"] >= 1).all()
    moved = np.abs(b["xtraj"] - a["xtraj"]).max(axis=(1, 2))
    assert (moved[early & (a["exit_code"] == 1)] > 0).all()                # they iterated: the trajectory changed
    # without the flag the early slots are left as they are (the documented keep-call behaviour)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve_iterations(10); a2 = s.get()
    s.solve_iterations(10, keep_iterate=True, keep_multipliers=True); c = s.get()
    np.testing.assert_array_equal(c["xtraj"][early], a2["xtraj"][early])
    s.close()


def test_grown_batch_starts_new_slots_fresh_and_evaluation_call_keeps_statistics():
    """Advisor (round 2): (i) state arrays are zeroed and a batch larger than any before starts its new slots like fresh capsules
    whatever the keep-flags say; (ii) an n_iter = 0 'complete' call keeps the iteration statistics of the iterations before it."""
    from mpc_planner_amd import scenes
    sc = scenes.make_scene(11, N=20, M=8, B=32)
    s = _solver("wave", 32, N=20, S=5, n_lin=8, M=8)
    s.set_batch(sc["xinit"][:8], sc["x0"][:8], sc["params"][:8]); s.solve_iterations(3)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.solve_iterations(3, keep_iterate=True, keep_multipliers=True); grown = s.get()       # slots 8.. have no state: fresh
    f = _solver("wave", 32, N=20, S=5, n_lin=8, M=8)
    f.set_batch(sc["xinit"], sc["x0"], sc["params"]); f.solve_iterations(3); fresh = f.get()
    for k in ("xtraj", "utraj", "pobj", "exit_code", "qp_status"):
        np.testing.assert_array_equal(grown[k][8:], fresh[k][8:], err_msg=k)
    assert np.isfinite(grown["xtraj"]).all()
    # (ii)
    f.solve_iterations(0, keep_iterate=True, keep_multipliers=True, complete=True); ev = f.get()
    np.testing.assert_array_equal(ev["sqp_iter"], fresh["sqp_iter"]); np.testing.assert_array_equal(ev["qp_iter_total"], fresh["qp_iter_total"])
    np.testing.assert_array_equal(ev["xtraj"], fresh["xtraj"])
    s.close(); f.close()


def test_slot_map_keys_the_state_by_slot_not_by_batch_position():
    """tmpc_set_slots (round-2 verdict: per-caller batches with one state slot per Solver): a Solver keeps ITS multipliers whatever
    position it has in a tick's batch and whatever subset of the Solvers the tick launches."""
    from mpc_planner_amd import scenes
    sc = scenes.make_scene(5, N=20, M=8, B=16)
    # reference: every trajectory alone on the identity map, two ticks with carried multipliers
    r = _solver("wave", 16, N=20, S=5, n_lin=8, M=8)
    r.set_batch(sc["xinit"], sc["x0"], sc["params"]); r.solve_iterations(10, keep_multipliers=True, new_solve=True); t1 = r.get()
    r.solve_iterations(10, keep_multipliers=True, new_solve=True); t2 = r.get()
    assert np.abs(t2["xtraj"] - t1["xtraj"]).max() > 1e-9                     # carried multipliers matter on these scenes
    # same Solvers through a slot map: tick 1 launches all 16 in a permuted order, tick 2 only a subset, again permuted
    s = _solver("wave", 16, N=20, S=5, n_lin=8, M=8)
    perm = np.random.default_rng(3).permutation(16)
    s.set_batch(sc["xinit"][perm], sc["x0"][perm], sc["params"][perm]); s.set_slots(perm)     # batch entry b is Solver perm[b]
    s.solve_iterations(10, keep_multipliers=True, new_solve=True); a = s.get()
    np.testing.assert_array_equal(a["xtraj"], t1["xtraj"][perm])
    sub = np.array([11, 2, 7, 14, 0])
    s.set_batch(sc["xinit"][sub], sc["x0"][sub], sc["params"][sub]); s.set_slots(sub)
    s.solve_iterations(10, keep_multipliers=True, new_solve=True); b = s.get()
    np.testing.assert_array_equal(b["xtraj"], t2["xtraj"][sub])                # each got its OWN first-tick multipliers
    np.testing.assert_array_equal(b["pobj"], t2["pobj"][sub])
    # a slot nothing was stored in starts fresh whatever the keep-flags say; clearing the map restores entry b <-> slot b
    f = _solver("wave", 16, N=20, S=5, n_lin=8, M=8)
    f.set_batch(sc["xinit"][:4], sc["x0"][:4], sc["params"][:4]); f.set_slots([12, 13, 14, 15])
    f.solve_iterations(10, keep_iterate=True, keep_multipliers=True); c = f.get()
    np.testing.assert_array_equal(c["xtraj"], t1["xtraj"][:4])
    f.set_slots(None)
    f.solve_iterations(10, keep_multipliers=True); d = f.get()                   # slots 0..3: never stored -> fresh again
    np.testing.assert_array_equal(d["xtraj"], t1["xtraj"][:4])
    # copying the state to a larger handle keeps every slot (a caller that outgrew its handle)
    g = _solver("wave", 32, N=20, S=5, n_lin=8, M=8)
    g.copy_state_from(r)
    g.set_batch(sc["xinit"], sc["x0"], sc["params"])
    g.solve_iterations(10, keep_multipliers=True, new_solve=True); e = g.get()
    r.solve_iterations(10, keep_multipliers=True, new_solve=True); t3 = r.get()
    np.testing.assert_array_equal(e["xtraj"], t3["xtraj"])
    for x in (r, s, f, g):
        x.close()
