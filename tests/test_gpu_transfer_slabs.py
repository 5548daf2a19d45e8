"""Control-tick handles move their inputs, the slot map and their outputs through ONE device allocation each with a pinned host mirror (round 6:
tmpc_set_batch = one asynchronous H2D copy, tmpc_set_slots without a stream synchronisation, tmpc_get = one D2H copy; csrc/tmpc_capi.hip).  Larger handles
keep separate allocations and the runtime's own path for pageable memory.  Both must be the same function of the caller's arrays: bit for bit, for every
batch size up to B_max, with and without a slot map, when the caller reuses or overwrites its arrays right after the call, and when only some outputs are
asked for."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "sqp_iter", "res_eq", "qp_iter_total")


def _scene(B, scene=4):
    from mpc_planner_amd import scenes
    return scenes.make_scene(scene, B=B, N=20, M=8)


def _dims():
    from mpc_planner_amd import solver
    return solver.default_dims(N=20, S=5, n_lin=8, M=8)


def test_small_and_large_handles_agree_bitwise():
    from mpc_planner_amd import solver
    sc = _scene(64)
    small = solver.BatchedSolver(_dims(), B_max=64)             # inputs 1.4 MB, outputs 90 KB: slabs (inputs above 512 KB take the direct path INTO the slab)
    large = solver.BatchedSolver(_dims(), B_max=4096)           # 93 MB of inputs: separate allocations
    for B in (1, 5, 8, 23, 64):                                  # 5, 8, 23: through the pinned mirror; 64: the direct path
        res = []
        for s in (small, large):
            s.set_batch(sc["xinit"][:B], sc["x0"][:B], sc["params"][:B]); s.solve(); res.append(s.get())
        for k in KEYS:
            assert np.array_equal(res[0][k], res[1][k]), (B, k)
    small.close(); large.close()


def test_the_callers_arrays_may_change_right_after_the_call():
    """tmpc_set_batch / tmpc_set_slots return before the copy has run (pinned mirror): they must have taken their own copy of the caller's data."""
    from mpc_planner_amd import solver
    sc = _scene(8)
    ref = solver.BatchedSolver(_dims(), B_max=8)
    ref.set_batch(sc["xinit"], sc["x0"], sc["params"]); ref.solve(); want = ref.get(); ref.close()
    s = solver.BatchedSolver(_dims(), B_max=8)
    xi, x0, pa = sc["xinit"].copy(), sc["x0"].copy(), sc["params"].copy()
    s.set_batch(xi, x0, pa)
    xi[:] = np.nan; x0[:] = np.nan; pa[:] = np.nan              # the caller's buffers are gone
    s.solve(); got = s.get()
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    # two set_batch calls back to back: the second rewrites the mirror only after the first copy has left it; the last one wins
    s.set_batch(sc["xinit"][::-1].copy(), sc["x0"][::-1].copy(), sc["params"][::-1].copy())
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.solve(); got = s.get()
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    s.close()


def test_slot_map_through_the_mirror():
    """Capsule state per slot (tmpc_set_slots + tmpc_solve_iterations with kept multipliers): a tick-size handle and a large one, the same permuted slot map,
    three ticks -- bit for bit."""
    from mpc_planner_amd import solver
    sc = _scene(8)
    slots = np.array([5, 0, 7, 2, 6, 1, 4, 3], np.int32)
    out = []
    for B_max in (8, 4096):
        s = solver.BatchedSolver(_dims(), B_max=B_max)
        ticks = []
        for t in range(3):
            s.set_batch(sc["xinit"], sc["x0"], sc["params"])
            m = slots.copy(); s.set_slots(m); m[:] = -1         # (the map is copied by the call)
            s.solve_iterations(4, keep_multipliers=True, complete=True, new_solve=True)
            ticks.append(s.get())
        out.append(ticks); s.close()
    for t in range(3):
        for k in KEYS:
            assert np.array_equal(out[0][t][k], out[1][t][k]), (t, k)
    assert not np.array_equal(out[0][0]["xtraj"], out[0][1]["xtraj"]) or (out[0][0]["qp_iter_total"] != out[0][1]["qp_iter_total"]).any()   # the kept multipliers act


def test_partial_outputs():
    from mpc_planner_amd import solver
    sc = _scene(5)
    s = solver.BatchedSolver(_dims(), B_max=8)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); full = s.get()
    ec = np.zeros(5, np.int32); pobj = np.zeros(5)
    null = C.c_void_p(None)
    rc = s.lib.tmpc_get(s._h, null, null, pobj.ctypes.data_as(C.c_void_p), ec.ctypes.data_as(C.c_void_p), null, null, null, null)
    assert rc == 0 and np.array_equal(ec, full["exit_code"]) and np.array_equal(pobj, full["pobj"])
    s.close()
