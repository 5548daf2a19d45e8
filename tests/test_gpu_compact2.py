"""GPU: the two-wave compact kernels (tmpc_solve_compact_kernel<..., NTH = 128>, 22 <= N <= 32: four trajectories per CU where the fast
two-wave kernel holds two; csrc/tmpc_capi.hpp pick_compact2_kernel).  The library takes them for launches the fast kernel cannot hold
resident at once -- which is only legitimate because the two compute bit for bit the same: asserted here on every registered shape
(TMPC_COMPACT2_MIN_B=0 forces the compact kernel, TMPC_NO_COMPACT=1 the fast one; both are read when the handle is created), against
the oracle, for the one-iteration protocol, and for the launch-size rule itself (a trajectory's result does not depend on what else
is in the launch).  The same rule, the other way round, for the one-wave shapes (N <= 21): launches the fast one-wave kernel holds resident run on
it, larger ones on the compact kernel; every compact one-wave instantiation is compared with its fast twin here."""
import os

import numpy as np
import pytest

from test_gpu_parity import _compare

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("lab_library")]      # (TMPC_* kernel-selection overrides: the lab build of the library, tests/conftest.py)
FIELDS = ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "sqp_iter", "qp_iter_total", "res_eq")


def _solver(force, B_max, **pkw):
    """force: "compact" / "fast" / None (the library's own rule)."""
    from mpc_planner_amd import solver
    for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT2_MIN_B", "TMPC_NO_ONE_WAVE_N30"):
        os.environ.pop(k, None)
    if force == "compact":
        os.environ["TMPC_COMPACT2_MIN_B"] = "0"
    elif force == "fast":
        os.environ["TMPC_NO_COMPACT"] = "1"
    if (pkw.get("row_model") == 1 and pkw.get("n_lin") == 5 and pkw.get("M") == 5) or (pkw.get("n_lin") == 8 and pkw.get("M") == 8 and not pkw.get("n_slk") and pkw.get("N", 20) > 21):
        os.environ["TMPC_NO_ONE_WAVE_N30"] = "1"          # (round 6: these shapes run on one-wave kernels at two lanes per stage; their two-wave twins stay in the library and are tested here)
    try:
        return solver.BatchedSolver(solver.default_dims(**pkw), B_max=B_max)
    finally:
        for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT2_MIN_B", "TMPC_NO_ONE_WAVE_N30"):
            os.environ.pop(k, None)


SHAPES = {
    # name: (scene kwargs, dims kwargs, oracle kwargs or None)
    "cfg3_ca": (dict(N=30, M=8, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), dict(cost_model=1)),
    "cfg3_mpcc": (dict(N=30, M=8, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), {}),
    "n30_8_8": (dict(N=30, M=8), dict(N=30, S=5, n_lin=8, M=8), {}),
    "n30_12_12": (dict(N=30, M=12), dict(N=30, S=5, n_lin=12, M=12), {}),
    "jackal_gaussian": (dict(N=30, M=5, S=3, chance=True), dict(N=30, S=3, n_lin=5, M=5, row_model=1), None),
    "n30_runtime_10_10": (dict(N=30, M=10), dict(N=30, S=5, n_lin=10, M=10), {}),
    "n22_8_8": (dict(N=22, M=8), dict(N=22, S=5, n_lin=8, M=8), {}),
    "n32_8_8": (dict(N=32, M=8), dict(N=32, S=5, n_lin=8, M=8), {}),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_compact_two_wave_is_bitwise_the_fast_two_wave_kernel_and_matches_the_oracle(shape):
    import oracle_lib as O
    from mpc_planner_amd import scenes
    skw, dkw, okw = SHAPES[shape]
    B = 32
    sc = scenes.make_scene(41, B=B, **skw)
    out = {}
    for force in ("fast", "compact"):
        s = _solver(force, B, **dkw)
        info = s.kernel_info()
        assert ("compact two-wave variant" in info) == (force == "compact"), info
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); out[force] = s.get(); s.close()
    for k in FIELDS:
        np.testing.assert_array_equal(out["compact"][k], out["fast"][k], err_msg=k)
    if okw is None:                                                    # Gaussian rows: orc_problem_set_gaussian
        pb = O.problem(N=dkw["N"], S=dkw["S"], n_lin=dkw["n_lin"], M=0, n_gauss=dkw["M"])
    else:
        pb = O.problem(**{k: v for k, v in dkw.items() if k != "cost_model"}, **okw)
    xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
    _compare(out["compact"], xt, ut, info)
    assert (info["exit_code"] == 1).sum() >= B // 2


def test_launch_size_rule_and_independence_of_the_launch():
    """More trajectories than the fast kernel holds resident (two per CU) -> the compact kernel; at most that many -> the fast kernel.  Every
    trajectory's result is the same in either launch."""
    from mpc_planner_amd import scenes
    dkw = dict(N=30, S=5, n_lin=8, M=8)
    b = scenes.make_batch(range(50, 60), N=30, M=8, B=64)                # 640 trajectories
    s = _solver(None, 640, **dkw)
    info = s.kernel_info()
    assert "compact two-wave variant" in info and "launches of more than" in info, info
    min_b = int(info.split("launches of more than ")[1].split()[0])
    assert 0 < min_b < 640 and min_b % 2 == 0, info                      # two workgroups per CU x CUs (512 on an MI355X)
    s.set_batch(b["xinit"], b["x0"], b["params"]); s.solve(); big = s.get()
    parts = []
    for lo in range(0, 640, 320):                                        # two launches of 320 (<= min_b on a full MI355X): the fast kernel
        s.set_batch(b["xinit"][lo:lo + 320], b["x0"][lo:lo + 320], b["params"][lo:lo + 320]); s.solve(); parts.append(s.get())
    s.close()
    for k in FIELDS:
        np.testing.assert_array_equal(big[k], np.concatenate([p[k] for p in parts], 0), err_msg=k)
    assert (big["exit_code"] == 1).mean() > 0.9


def test_one_iteration_protocol_on_the_compact_two_wave_kernel():
    """10 x solveOneIteration == solve(), bitwise, with the persistent state (slots) written and read by the two-wave compact kernel."""
    from mpc_planner_amd import scenes
    import test_gpu_parity as T
    sc = T._make_infeasible(scenes.make_scene(7, N=30, M=8, B=48), [9, 33])
    s = _solver("compact", 48, N=30, S=5, n_lin=8, M=8, qp_iter_max=6)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); ref = s.get()
    assert (ref["exit_code"] != 1).any()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.solve_iterations(1, complete=False)
    for i in range(9):
        s.solve_iterations(1, keep_iterate=True, keep_multipliers=True, complete=(i == 8))
    g = s.get()
    for k in ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "res_eq"):
        np.testing.assert_array_equal(g[k], ref[k], err_msg=k)
    s.close()


@pytest.mark.parametrize("shape", ["cfg1", "cfg2", "cfg4", "cfg5", "runtime_7_rows_per_lane", "runtime_10_rows_per_lane"])
def test_one_wave_shapes_fast_kernel_for_resident_launches_is_bitwise_the_compact_kernel(shape):
    """N <= 21: launches the fast one-wave kernel holds resident at once (four per CU) run on it, larger ones on the compact kernel (eight per
    CU, persistent) -- legitimate only because the two compute bit for bit the same (TMPC_COMPACT_MIN_B=0: the compact kernel for every launch)."""
    from mpc_planner_amd import scenes, solver
    skw, dkw = {"cfg1": (dict(N=20, M=4, guidance=False), dict(N=20, S=5, n_lin=0, M=4)),
                "runtime_7_rows_per_lane": (dict(N=20, M=3), dict(N=20, S=5, n_lin=3, M=3)),
                "runtime_10_rows_per_lane": (dict(N=20, M=6), dict(N=20, S=5, n_lin=6, M=6)),
                "cfg2": (dict(N=20, M=8), dict(N=20, S=5, n_lin=8, M=8)),
                "cfg4": (dict(N=20, M=12), dict(N=20, S=5, n_lin=12, M=12)),
                "cfg5": (dict(N=20, M=8, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1))}[shape]
    B = 32
    sc = scenes.make_scene(43, B=B, **skw)
    out = {}
    for env in ("0", None):
        os.environ.pop("TMPC_COMPACT_MIN_B", None)
        if env is not None:
            os.environ["TMPC_COMPACT_MIN_B"] = env
        try:
            s = solver.BatchedSolver(solver.default_dims(**dkw), B_max=B)
        finally:
            os.environ.pop("TMPC_COMPACT_MIN_B", None)
        info = s.kernel_info()
        assert "compact" in info and (("launches of at most 0 " in info or "fast one-wave variant" not in info) == (env == "0")), info
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); out[env] = s.get(); s.close()
    for k in FIELDS:
        np.testing.assert_array_equal(out[None][k], out["0"][k], err_msg=k)
    assert (out[None]["exit_code"] == 1).sum() >= B // 2


def _forced_compact(B_max, **dkw):
    """The compact kernel of the shape for every launch size (one-wave: TMPC_COMPACT_MIN_B=0; two-wave: TMPC_COMPACT2_MIN_B=0)."""
    from mpc_planner_amd import solver
    os.environ["TMPC_COMPACT_MIN_B"] = "0"; os.environ["TMPC_COMPACT2_MIN_B"] = "0"
    try:
        return solver.BatchedSolver(solver.default_dims(**dkw), B_max=B_max)
    finally:
        os.environ.pop("TMPC_COMPACT_MIN_B", None); os.environ.pop("TMPC_COMPACT2_MIN_B", None)


def test_one_iteration_protocol_on_the_compact_one_wave_kernel():
    """Small launches run on the fast kernels since the launch-size rule: the persistent-state protocol on the COMPACT one-wave kernel (what a
    launch of thousands of solver slots uses) is exercised here by forcing it."""
    from mpc_planner_amd import scenes
    import test_gpu_parity as T
    sc = T._make_infeasible(scenes.make_scene(7, N=20, M=8, B=64), [9, 33])
    s = _forced_compact(64, N=20, S=5, n_lin=8, M=8, qp_iter_max=6)
    assert "fast one-wave variant" not in s.kernel_info()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); ref = s.get()
    assert (ref["sqp_iter"] < 10).any() and (ref["exit_code"] != 1).any()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    s.solve_iterations(1, complete=False)
    for i in range(9):
        s.solve_iterations(1, keep_iterate=True, keep_multipliers=True, complete=(i == 8))
    g = s.get()
    for k in ("xtraj", "utraj", "pobj", "exit_code", "qp_status", "res_eq"):
        np.testing.assert_array_equal(g[k], ref[k], err_msg=k)
    s.close()


@pytest.mark.parametrize("N", [20, 30])
def test_param_sharing_hint_on_the_compact_kernels(N):
    """tmpc_set_param_sharing on the compact kernels (one-wave N = 20, two-wave N = 30), forced for a two-set launch: bitwise neutral."""
    from mpc_planner_amd import scenes, solver as S
    b = scenes.make_batch(range(60, 62), N=N, M=8, B=64)
    s = _forced_compact(128, N=N, S=5, n_lin=8, M=8)
    base = S.param_sharing_map(b["params"], s.dims, 64)
    assert (base != np.arange(128)).sum() == 126
    s.set_batch(b["xinit"], b["x0"], b["params"]); s.solve(); ref = s.get()
    s.set_param_sharing(base); s.solve(); got = s.get()
    for k in FIELDS:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert (ref["exit_code"] == 1).mean() > 0.9
    s.close()


ONE_WAVE_N30 = {
    "jackal default (5 + 5 Gaussian rows, N = 30)": (dict(N=30, S=3, n_lin=5, M=5, row_model=1), dict(N=30, M=5, S=3, chance=True), dict(N=30, S=3, n_lin=5, M=0, n_gauss=5)),
    "jackalsimulator stack at its shipped horizon (8 + 8, N = 30)": (dict(N=30, S=5, n_lin=8, M=8), dict(N=30, M=8), dict(N=30, S=5, n_lin=8, M=8)),
    "8 + 8, N = 22": (dict(N=22, S=5, n_lin=8, M=8), dict(N=22, M=8), dict(N=22, S=5, n_lin=8, M=8)),
    "8 + 8, N = 32 (every lane of the wave in use)": (dict(N=32, S=5, n_lin=8, M=8), dict(N=32, M=8), dict(N=32, S=5, n_lin=8, M=8)),
}


@pytest.mark.parametrize("name", sorted(ONE_WAVE_N30))
def test_one_wave_kernels_at_two_lanes_per_stage(name):
    """Round 6: mpc_planner_jackal's default stack (N = 30, 5 topology + 5 Gaussian rows: generate_jackal_solver.py:53-73) and the jackalsimulator stack at its shipped
    horizon run on ONE wave per trajectory at two lanes per stage -- the compact kernel for launches beyond the fast one-wave kernel's resident set, the fast one
    below: bit for bit the same, and the oracle's integers; the product library's own rule picks them (no switch)."""
    import oracle_lib as O
    from mpc_planner_amd import scenes, solver
    dkw, skw, okw = ONE_WAVE_N30[name]
    B = 32
    sc = scenes.make_scene(41, B=B, **skw)
    out = {}
    for force in ("fast", "compact"):
        for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT_MIN_B"):
            os.environ.pop(k, None)
        os.environ["TMPC_NO_COMPACT" if force == "fast" else "TMPC_COMPACT_MIN_B"] = "1" if force == "fast" else "0"
        try:
            s = solver.BatchedSolver(solver.default_dims(**dkw), B_max=B)
        finally:
            os.environ.pop("TMPC_NO_COMPACT", None); os.environ.pop("TMPC_COMPACT_MIN_B", None)
        info = s.kernel_info()
        assert "two waves per trajectory" not in info.split(";")[0], info
        assert ("compact" in info.split(";")[0]) == (force == "compact"), info
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); out[force] = s.get(); s.close()
    for k in FIELDS:
        np.testing.assert_array_equal(out["compact"][k], out["fast"][k], err_msg=k)
    pb = O.problem(**okw)
    xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
    _compare(out["compact"], xt, ut, info)
    assert (info["exit_code"] == 1).sum() >= B // 2
    # the library's own rule: a saturated launch takes the compact kernel, a tick-size one the fast kernel -- same results for a trajectory in either
    s = solver.BatchedSolver(solver.default_dims(**dkw), B_max=4096)
    big = [np.tile(sc[k], (128,) + (1,) * (sc[k].ndim - 1)) for k in ("xinit", "x0", "params")]
    s.set_batch(*big); s.solve(); g_big = s.get()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); g_small = s.get(); s.close()
    for k in FIELDS:
        np.testing.assert_array_equal(g_big[k][:B], g_small[k], err_msg=k)
        np.testing.assert_array_equal(g_small[k], out["fast"][k], err_msg=k)
