"""Latency mode 3 (round 6; round-5 verdict next-1): FOUR waves per trajectory -- the control-tick kernel for launches that are one dependent chain
deep (guidance_constraints.cpp:279-361 at its deployed size: 4 + 1 planners; configs[1]'s 64-trajectory tick; cfg 4's share of an 8-GPU split; cfg 5).
Every instantiation against the oracle: exit codes, QP status, SQP and interior-point iteration counts bit-exact, trajectories <= 1e-8; and
bitwise reproducible from launch to launch (every LDS accumulation has a fixed order, also across the four waves)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(dims_kw, scene_kw, B, scene=5, orc_kw=None, mode=3):
    import oracle_lib as O
    from mpc_planner_amd import scenes, solver
    sc = scenes.make_scene(scene, B=B, **scene_kw)
    n = sc["xinit"].shape[0]
    s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=n)
    assert s.set_latency_mode(mode), "no four-wave variant for this shape"
    assert s.latency_mode_capacity(mode) == 256                      # one workgroup (four waves) per CU: what the variant serves well (MI355X: 256 CUs)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); g = s.get()
    s.solve(); g2 = s.get()
    best = s.select_best()
    s.close()
    for k in ("xtraj", "utraj", "pobj"):
        assert np.array_equal(g[k], g2[k]), k                        # launch-to-launch: bit for bit
    pb = O.problem(**(orc_kw or dims_kw))
    xt, ut, o = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(n, -1), sc["params"].reshape(n, -1))
    assert (g["exit_code"] == o["exit_code"]).all() and (g["sqp_iter"] == o["sqp_iter"]).all()
    ok = o["exit_code"] == 1
    assert ok.any()
    assert (g["qp_status"][ok] == o["qp_status"][ok]).all() and (g["qp_iter_total"][ok] == o["qp_iter_total"][ok]).all()
    sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0); su = np.maximum(np.abs(ut[ok]).max(axis=2, keepdims=True), 1.0)
    ex = (np.abs(g["xtraj"][ok] - xt[ok]) / sx).max(); eu = (np.abs(g["utraj"][ok] - ut[ok]) / su).max()
    assert ex < 1e-4 and eu < 1e-4                                   # the contract
    assert ex < 1e-8 and eu < 1e-8, (ex, eu)                          # the regression line
    assert best == O.find_best(g["pobj"], g["exit_code"])
    print(f"[quad] {dims_kw} B {n}: {max(ex, eu):.2e}")
    return g


@pytest.mark.parametrize("scene", [0, 3, 7])
def test_tick_of_64_matches_oracle(scene):
    _run(dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8), 64, scene)


def test_deployed_size_4_plus_1_planners():
    """4 guidance planners + the non-guided T-MPC++ planner (guidance_planner.yaml:11, guidance_constraints.cpp:40-52)."""
    _run(dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8, tmpc_pp=True), 4)


def test_cfg1_cfg4_cfg5_shapes_on_the_runtime_shape_instantiation():
    _run(dict(N=20, S=5, n_lin=0, M=4), dict(N=20, M=4, guidance=False), 1)
    _run(dict(N=20, S=5, n_lin=12, M=12), dict(N=20, M=12), 96)
    _run(dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), dict(N=20, M=8, slack=True, n_scenario=24), 32)


def test_shorter_horizons():
    """N < 20: node N sits on a regular stage slot of the four-wave factorisation instead of the spare lanes."""
    for N in (2, 5, 11, 16, 19):
        _run(dict(N=N, S=5, n_lin=8, M=8), dict(N=N, M=8), 8)


def test_equal_to_mode_2_to_rounding():
    from mpc_planner_amd import scenes, solver
    sc = scenes.make_scene(2, N=20, M=8, B=64)
    res = {}
    for mode in (2, 3):
        s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8), B_max=64)
        assert s.set_latency_mode(mode)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); res[mode] = s.get(); s.close()
    assert (res[2]["exit_code"] == res[3]["exit_code"]).all() and (res[2]["qp_iter_total"] == res[3]["qp_iter_total"]).all()
    ok = res[2]["exit_code"] == 1
    assert np.abs(res[2]["xtraj"][ok] - res[3]["xtraj"][ok]).max() < 1e-8


def test_no_four_wave_variant_beyond_n_31():
    from mpc_planner_amd import solver
    s = solver.BatchedSolver(solver.default_dims(N=32, S=5, n_lin=8, M=8), B_max=8)
    assert not s.set_latency_mode(3) and s.latency_mode_capacity(3) == 0
    s.close()


# ---- 21 <= N <= 31: the horizon the reference ships (mpc_planner_jackalsimulator/config/settings.yaml N: 30, mpc_planner_jackal likewise) ----------------
WIDE = {
    "cfg3 (rosnavigation stack: slack model, decomp rows)": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), dict(N=30, M=8, slack=True, n_decomp=12), 16, None),
    "cfg3 with the curvature-aware cost": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), dict(N=30, M=8, slack=True, n_decomp=12), 16, None),
    "jackalsimulator stack at its shipped horizon (8 + 8 rows, N = 30)": (dict(N=30, S=5, n_lin=8, M=8), dict(N=30, M=8), 16, None),
    "12 + 12 rows, N = 30": (dict(N=30, S=5, n_lin=12, M=12), dict(N=30, M=12), 16, None),
    "scenario rows, slack, N = 30": (dict(N=30, S=5, n_lin=0, M=0, n_slk=24, slack=1), dict(N=30, M=8, slack=True, n_scenario=24), 8, None),
}


@pytest.mark.parametrize("name", sorted(WIDE))
def test_shipped_horizon_n30_matches_oracle(name):
    dims_kw, scene_kw, B, orc_kw = WIDE[name]
    for scene in (1, 6):
        _run(dims_kw, scene_kw, B, scene, orc_kw)


def test_deployed_size_4_plus_1_planners_at_the_shipped_horizon():
    _run(dict(N=30, S=5, n_lin=8, M=8), dict(N=30, M=8, tmpc_pp=True), 4)


def test_horizons_between_21_and_31():
    """Every lane map of the wide variant: N = 21 is the first horizon on it, N = 31 fills the 32 node slots; odd / even block counts per reduction level."""
    for N in (21, 22, 25, 28, 31):
        _run(dict(N=N, S=5, n_lin=8, M=8), dict(N=N, M=8), 6)


def test_wide_variant_equal_to_mode_2_to_rounding():
    from mpc_planner_amd import scenes, solver
    sc = scenes.make_scene(2, N=30, M=8, B=16, slack=True, n_decomp=12)
    n = sc["xinit"].shape[0]
    res = {}
    for mode in (2, 3):
        s = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), B_max=n)
        assert s.set_latency_mode(mode)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); res[mode] = s.get(); s.close()
    assert (res[2]["exit_code"] == res[3]["exit_code"]).all() and (res[2]["qp_iter_total"] == res[3]["qp_iter_total"]).all()
    ok = res[2]["exit_code"] == 1
    assert np.abs(res[2]["xtraj"][ok] - res[3]["xtraj"][ok]).max() < 1e-8
