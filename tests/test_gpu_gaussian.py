"""GPU parity of the hand-written Gaussian chance-constraint rows (tmpc_dims::row_model = 1; GaussianConstraintModule,
gaussian_constraints.py:33-113 -- the collision-avoidance submodule of mpc_planner_jackal's default T-MPC, generate_jackal_solver.py:53-73)
through the C-ABI: device stage functions vs golden vectors made by executing the reference's own script
(tests/golden/make_golden_gaussian.py), and solves vs the CPU oracle (orc_problem_set_gaussian) on the kernels that carry the rows:
runtime-shape two-wave (N = 30, the jackal default), runtime-shape one-wave (N = 20), generic (more rows than those hold).  The same stack as a GENERATED library
is tests/test_gpu_parity.py::test_generated_jackal_default_solver_matches_oracle; this is the zero-scratch hand-written path to it.
Tolerances as in test_gpu_parity.py: integer work bit-exact, trajectories < 1e-4 relative per stage (observed far below)."""
import json
import os

import numpy as np
import pytest

from test_gpu_parity import _check_selection, _compare, _solver

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_gaussian_stage_functions_match_reference_golden():
    with open(os.path.join(HERE, "golden", "stage_functions_gaussian.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        s = _solver(N=case["N"], S=case["S"], n_lin=case["M"], M=case["M"], row_model=1, B_max=4)
        assert s.npar == case["npar"] == 82
        o = s.debug_eval_stage(case["z"], case["p"])
        np.testing.assert_allclose(o["cost"][0], case["cost"], rtol=1e-11)
        np.testing.assert_allclose(o["h"][0], case["h"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(o["h_jac"][0], case["h_jac"], rtol=1e-9, atol=1e-11)
        # second derivatives of the rows: the Lagrangian Hessian with multipliers on the rows only, minus the cost's share
        rng = np.random.default_rng(3)
        lam = rng.normal(size=case["nh"])
        o0 = s.debug_eval_stage(case["z"], case["p"], pi=np.zeros(5), lamh=np.zeros(case["nh"]))
        o2 = s.debug_eval_stage(case["z"], case["p"], pi=np.zeros(5), lamh=lam)
        ref = np.tensordot(lam, np.array(case["h_hess"]), 1)
        np.testing.assert_allclose(o2["lag_hess"][0] - o0["lag_hess"][0], ref, rtol=1e-8, atol=1e-10)
        assert np.abs(ref).max() > 1e-3                                   # (the Gaussian rows are curved: it is not a comparison of zeros)
        s.close()


GAUSS_SHAPES = {
    # mpc_planner_jackal's default: N = 30, 3 segments, 5 topology + 5 Gaussian rows -> runtime-shape two-wave kernel <-1, 6, 4, 128, CM = 2>
    "jackal_two_wave": (dict(N=30, M=5, S=3, chance=True), dict(N=30, S=3, n_lin=5, M=5), 16, (1, 2, 3, 4)),
    # N = 20, 8 + 8 rows -> runtime-shape one-wave kernel <-1, 13, 3, 64, CM = 2>
    "n20_one_wave": (dict(N=20, M=8, chance=True), dict(N=20, n_lin=8, M=8), 32, (0, 3)),
    # N = 30 with 12 + 12 rows -> runtime-shape two-wave kernel <-1, 12, 4, 128, CM = 2>
    "n30_12_rows": (dict(N=30, M=12, chance=True), dict(N=30, n_lin=12, M=12), 16, (1, 5)),
    # more rows than the one-wave kernel holds (14 + 14 + 14 = 42 per stage) -> generic kernel <CM = 2>
    "n20_generic": (dict(N=20, M=14, chance=True), dict(N=20, n_lin=14, M=14), 16, (2,)),
}


@pytest.mark.parametrize("shape", sorted(GAUSS_SHAPES))
def test_gaussian_solve_matches_oracle(shape):
    import oracle_lib as O
    from mpc_planner_amd import scenes
    skw, pkw, B, scene_ids = GAUSS_SHAPES[shape]
    pkw = dict(dict(S=5), **pkw)
    s = _solver(B_max=B, row_model=1, **pkw)
    info_txt = s.kernel_info()
    assert ("generic" in info_txt) == (shape == "n20_generic"), info_txt
    pb = O.problem(N=pkw["N"], S=pkw["S"], n_lin=pkw["n_lin"], M=0, n_gauss=pkw["M"])
    assert pb.npar == s.npar
    n_ok = 0
    for scene in scene_ids:
        sc = scenes.make_scene(scene, B=B, **skw)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); got = s.get()
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        _compare(got, xt, ut, info)
        _check_selection(s.select_best(), got, info)
        n_ok += int((info["exit_code"] == 1).sum())
    assert n_ok >= 0.5 * B * len(scene_ids), n_ok
    s.close()


def test_gaussian_rows_are_refused_where_they_do_not_exist():
    from mpc_planner_amd import solver
    s = _solver(B_max=4, row_model=1)
    with pytest.raises(solver.TmpcError):
        s.set_throughput_mode(True)                              # the lane kernels have ellipsoid rows only
    assert s.set_latency_mode(2) and s.set_latency_mode(3)      # since round 6 the latency variants carry the Gaussian rows (run-time shapes)
    s.close()
    with pytest.raises(solver.TmpcError):
        _solver(B_max=4, row_model=2)


@pytest.mark.parametrize("N,mode", [(20, 0), (30, 0), (30, 3)])
def test_gaussian_rows_with_the_curvature_aware_cost(N, mode):
    """Round-5 verdict missing-7, first half: CurvatureAwareContouringModule (curvature_aware_contouring.py:48-105) together with GaussianConstraintModule
    (gaussian_constraints.py:68-117) -- stage model CM = 3: the generic kernel (any N), and the four-wave tick kernel at the shipped horizon N = 30 -- against
    the oracle (cost_model = 1, n_gauss): every integer, 1e-8."""
    import oracle_lib as O
    from mpc_planner_amd import scenes
    B = 16
    s = _solver(B_max=B, N=N, S=3, n_lin=5, M=5, row_model=1, cost_model=1)
    assert "generic" in s.kernel_info()                                  # the default kernel of the combination
    if mode:
        assert s.set_latency_mode(mode)
    else:
        assert not s.set_latency_mode(1) and not s.set_latency_mode(2)
        s.set_latency_mode(0)
    pb = O.problem(N=N, S=3, n_lin=5, M=0, n_gauss=5, cost_model=1)
    mpcc = O.problem(N=N, S=3, n_lin=5, M=0, n_gauss=5)                  # the same rows under the MPCC cost: another problem
    assert pb.npar == s.npar
    n_ok, moved = 0, 0.0
    for scene in (1, 4):
        sc = scenes.make_scene(scene, B=B, N=N, M=5, S=3, chance=True)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); got = s.get()
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        _compare(got, xt, ut, info)
        n_ok += int((info["exit_code"] == 1).sum())
        xm, _, im = O.solve_batch(mpcc, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        both = (info["exit_code"] == 1) & (im["exit_code"] == 1)
        moved = max(moved, float(np.abs(xt[both] - xm[both]).max()))
    assert n_ok >= B and moved > 1e-3, (n_ok, moved)                    # (a kernel that ignored the cost flag would match `mpcc`, not `pb`)
    s.close()


@pytest.mark.parametrize("shape,mode", [("jackal_two_wave", 2), ("jackal_two_wave", 3), ("n20_one_wave", 2), ("n20_one_wave", 3)])
def test_gaussian_rows_on_the_latency_variants(shape, mode):
    """Round-5 verdict next-7: mpc_planner_jackal's shipped default stack (generate_jackal_solver.py:53-73, gaussian_constraints.py:68-117) on the tick
    kernels -- latency modes 2 and 3 (four waves) at N = 30 (the default's horizon) and N = 20 -- against the oracle: every integer, 1e-8."""
    import oracle_lib as O
    from mpc_planner_amd import scenes
    skw, pkw, B, scene_ids = GAUSS_SHAPES[shape]
    pkw = dict(dict(S=5), **pkw)
    s = _solver(B_max=B, row_model=1, **pkw)
    assert s.set_latency_mode(mode)
    pb = O.problem(N=pkw["N"], S=pkw["S"], n_lin=pkw["n_lin"], M=0, n_gauss=pkw["M"])
    for scene in scene_ids:
        sc = scenes.make_scene(scene, B=B, **skw)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); got = s.get()
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        _compare(got, xt, ut, info)
    s.close()


def test_gaussian_rows_on_the_one_wave_compact_kernel():
    """... and on the throughput side at N <= 20: a launch beyond the fast kernel's resident set runs the compact (two waves per SIMD, persistent)
    run-time-shape kernel with CM = 2 -- bitwise the fast kernel's results."""
    from mpc_planner_amd import scenes
    pkw = dict(N=20, S=5, n_lin=5, M=5)
    sc = scenes.make_scene(3, B=32, N=20, M=5, chance=True)
    rep = 48                                                            # 1536 trajectories > 4 per CU x 256 CUs
    big = [np.tile(sc[k], (rep,) + (1,) * (sc[k].ndim - 1)) for k in ("xinit", "x0", "params")]
    s = _solver(B_max=32 * rep, row_model=1, **pkw)
    assert "compact" in s.kernel_info(), s.kernel_info()
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); small = s.get()       # the fast kernel (launch within its resident set)
    s.set_batch(*big); s.solve(); large = s.get()                                        # the compact kernel
    for k in ("exit_code", "sqp_iter", "qp_iter_total", "xtraj", "utraj", "pobj"):
        assert np.array_equal(np.tile(small[k], (rep,) + (1,) * (small[k].ndim - 1)), large[k]), k
    assert (small["exit_code"] == 1).mean() > 0.5
    s.close()
