"""GPU parity of the curvature-aware contouring configuration (BASELINE configs[2] "Jackal CA-MPC + decomp_util static constraints";
tmpc_dims::cost_model = 1) through the C-ABI: device stage functions vs golden vectors made by executing the reference's own
curvature_aware_contouring.py (tests/golden/make_golden_ca.py), and solves vs the CPU oracle (orc_problem.cost_model = 1) on the three
kernels that carry the cost: two-wave (20,8,4) at N = 30 (the cfg-3 shape), the runtime-shape one-wave kernel at N = 20, the generic kernel.
Tolerances as in test_gpu_parity.py: integer work bit-exact, trajectories < 1e-4 relative per stage (observed far below)."""
import json
import os

import numpy as np
import pytest

from test_gpu_parity import _check_selection, _compare, _solver

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_ca_stage_functions_match_reference_golden():
    with open(os.path.join(HERE, "golden", "stage_functions_ca.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        s = _solver(N=case["N"], n_lin=case["n_lin"], M=case["M"], n_slk=case["n_dec"], slack=case["slack"], cost_model=1, B_max=4)
        assert s.npar == case["npar"]
        o = s.debug_eval_stage(case["z"], case["p"])
        H = np.array(case["cost_hess"])
        np.testing.assert_allclose(o["cost"][0], case["cost"], rtol=1e-11)
        np.testing.assert_allclose(o["cost_grad"][0], case["cost_grad"][:7], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(o["cost_hess"][0], H[:7, :7], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(o["h"][0], case["h"], rtol=1e-11, atol=1e-12)
        assert abs(o["cost_hess"][0][4, 6]) > 0 and abs(o["cost_hess"][0][5, 2]) > 0       # psi / v couple with (x, y, s): the full 7 x 7 MIRROR
        # the Lagrangian Hessian is symmetric and MIRROR gives the eigenvalue-mirrored matrix of it (coupled branch of mirror7)
        rng = np.random.default_rng(11)
        pi = rng.normal(size=5); lam = rng.normal(size=case["nh"]) * 0.1
        o2 = s.debug_eval_stage(case["z"], case["p"], pi=pi, lamh=lam)
        W = o2["lag_hess"][0]
        np.testing.assert_allclose(W, W.T, rtol=0, atol=1e-12)
        e, V = np.linalg.eigh(W)
        e2 = np.where(np.abs(e) <= 1e-4, 1e-4, np.abs(e))
        np.testing.assert_allclose(o2["mirror"][0], (V * e2) @ V.T, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(e).max()))
        s.close()


CA_SHAPES = {
    # BASELINE configs[2]: slack model + guidance + 8 ellipsoids + 12 decomp rows, N = 30 -> two-wave kernel <20, 8, 4, 128, CM = 1>
    "cfg3_two_wave": (dict(N=30, M=8, slack=True, n_decomp=12), dict(N=30, n_lin=8, M=8, n_slk=12, slack=1), 16, (1, 2)),
    # the CA cost on the plain unicycle T-MPC stack, N = 20 -> runtime-shape one-wave kernel <-1, 13, 3, 64, CM = 1>
    "n20_one_wave": (dict(N=20, M=8), dict(N=20, n_lin=8, M=8), 32, (0, 3)),
    # more rows than the one-wave kernel holds (12 + 12 + 12 + 14 = 50 per stage) -> generic kernel <CM = 1>
    "n20_generic": (dict(N=20, M=12, S=8, slack=True, n_decomp=12), dict(N=20, S=8, n_lin=12, M=12, n_slk=12, slack=1), 16, (1,)),
}


@pytest.mark.parametrize("shape", sorted(CA_SHAPES))
def test_ca_solve_matches_oracle(shape):
    import oracle_lib as O
    from mpc_planner_amd import scenes
    skw, pkw, B, scene_ids = CA_SHAPES[shape]
    pkw = dict(dict(S=5), **pkw)
    s = _solver(B_max=B, cost_model=1, **pkw)
    info_txt = s.kernel_info()
    assert ("generic" in info_txt) == (shape == "n20_generic"), info_txt
    pb = O.problem(cost_model=1, **pkw)
    pb0 = O.problem(**pkw)
    n_ok = 0
    for scene in scene_ids:
        sc = scenes.make_scene(scene, B=B, **skw)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); got = s.get()
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        _compare(got, xt, ut, info)
        _check_selection(s.select_best(), got, info)
        n_ok += int((info["exit_code"] == 1).sum())
        # it IS another problem than the MPCC stack's on the same parameters
        xt0, _, info0 = O.solve_batch(pb0, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        both = (info["exit_code"] == 1) & (info0["exit_code"] == 1)
        assert both.any() and np.abs(xt[both] - xt0[both]).max() > 1e-3
    assert n_ok >= 0.9 * B * len(scene_ids), n_ok
    s.close()


def test_ca_baseline_size_512_trajectories():
    """configs[2] at the size it names: 512 trajectories on one GPU (8 ticks x 64), every one compared with the oracle."""
    import oracle_lib as O
    from mpc_planner_amd import scenes
    pkw = dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1)
    sc = scenes.make_batch(range(0, 8), N=30, M=8, B=64, slack=True, n_decomp=12)
    B = sc["xinit"].shape[0]
    s = _solver(B_max=B, cost_model=1, **pkw)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); got = s.get()
    xt, ut, info = O.solve_batch(O.problem(cost_model=1, **pkw), sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
    _compare(got, xt, ut, info)
    assert (info["exit_code"] == 1).mean() >= 0.95, (info["exit_code"] == 1).mean()
    assert np.all(got["xtraj"][:, :, 5] == 0.0)                 # the pinned slack state
    s.close()


def test_ca_is_refused_where_it_does_not_exist():
    from mpc_planner_amd import solver
    s = _solver(B_max=4, cost_model=1)
    with pytest.raises(solver.TmpcError):
        s.set_throughput_mode(True)                              # the lane kernels carry the MPCC cost only
    assert not s.set_latency_mode(2)                             # no parallel-in-time variant: accepted, runs the default kernel
    s.close()
    with pytest.raises(solver.TmpcError):
        _solver(B_max=4, cost_model=2)
