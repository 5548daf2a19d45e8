"""tests/golden/solve_fixtures.json (inputs + outputs of whole solves, made by tests/golden/make_solve_fixtures.py; replayable
through the reference's own acados solver with tools/acados_replay.py): the oracle still reproduces them (CPU), and the HIP path
-- wave-per-trajectory and lane-per-trajectory kernels -- reproduces them on the GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, "golden", "solve_fixtures.json")) as fh:
        return json.load(fh)["cases"]


def test_oracle_reproduces_fixtures():
    import oracle_lib as O
    for c in _cases():
        pb = O.problem(**c["problem"])
        N, nx, nvar = c["N"], c["nx"], c["nvar"]
        xt, ut, info = O.solve(pb, np.array(c["xinit"]), np.array(c["x0"]), np.array(c["params"]))
        ref = c["oracle_qp_tol_1e_5"]
        assert info.exit_code == ref["exit_code"] and info.sqp_iter == ref["sqp_iter"] and info.qp_iter_total == ref["qp_iter_total"]
        np.testing.assert_allclose(xt.ravel(), ref["xtraj"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(ut.ravel(), ref["utraj"], rtol=0, atol=1e-9)
        # and the three stored outputs relate as tests/test_independent_rti.py says
        a = np.array(c["active_set_rti"]["xtraj"]); t9 = np.array(c["oracle_qp_tol_1e_9"]["xtraj"]); t8 = np.array(c["oracle_qp_tol_1e_8"]["xtraj"])
        assert np.abs(a - t9).max() < 1e-6 and np.abs(a - np.array(ref["xtraj"])).max() < 5e-3
        assert np.abs(t8 - t9).max() < 1e-5 and np.abs(a - t8).max() < 1e-5      # 1e-8: the tolerance the acados hand-off runs at (tools/acados_replay.py)
        x8, u8, i8 = O.solve(O.problem(qp_tol=1e-8, **c["problem"]), np.array(c["xinit"]), np.array(c["x0"]), np.array(c["params"]))
        r8 = c["oracle_qp_tol_1e_8"]
        assert i8.exit_code == r8["exit_code"] and i8.sqp_iter == r8["sqp_iter"] and i8.qp_iter_total == r8["qp_iter_total"]
        np.testing.assert_allclose(x8.ravel(), r8["xtraj"], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["wave", pytest.param("lanes", marks=pytest.mark.lanes)])
def test_hip_path_reproduces_fixtures(mode):
    from mpc_planner_amd import solver
    for c in _cases():
        dims = solver.default_dims(**c["problem"])
        s = solver.BatchedSolver(dims, B_max=1)
        if mode == "lanes":
            s.set_throughput_mode(True)
        N, nx, nvar = c["N"], c["nx"], c["nvar"]
        s.set_batch(np.array(c["xinit"]).reshape(1, nx), np.array(c["x0"]).reshape(1, N + 1, nvar), np.array(c["params"]).reshape(1, N, -1))
        s.solve(); g = s.get(); s.close()
        ref = c["oracle_qp_tol_1e_5"]
        assert g["exit_code"][0] == ref["exit_code"] and g["sqp_iter"][0] == ref["sqp_iter"] and g["qp_iter_total"][0] == ref["qp_iter_total"]
        xr = np.array(ref["xtraj"]).reshape(N + 1, nx); ur = np.array(ref["utraj"]).reshape(N, 2)
        sx = np.maximum(np.abs(xr).max(axis=1, keepdims=True), 1.0); su = np.maximum(np.abs(ur).max(axis=1, keepdims=True), 1.0)
        assert (np.abs(g["xtraj"][0] - xr) / sx).max() < 1e-8 and (np.abs(g["utraj"][0] - ur) / su).max() < 1e-8      # the regression line of tests/test_gpu_parity.py
        assert abs(g["pobj"][0] - ref["pobj"]) < 1e-8 * max(1.0, abs(ref["pobj"]))
        if mode == "wave":                              # ... and at the hand-off tolerance, in BOTH Riccati forms (the fixture is the oracle's square-root form)
            for form in (0, 1):
                s8 = solver.BatchedSolver(solver.default_dims(**c["problem"], qp_tol=1e-8, riccati_form=form), B_max=1)
                s8.set_batch(np.array(c["xinit"]).reshape(1, nx), np.array(c["x0"]).reshape(1, N + 1, nvar), np.array(c["params"]).reshape(1, N, -1))
                s8.solve(); g8 = s8.get(); s8.close()
                r8 = c["oracle_qp_tol_1e_8"]
                assert g8["exit_code"][0] == r8["exit_code"] and g8["sqp_iter"][0] == r8["sqp_iter"], (c["config"], form)
                x8 = np.array(r8["xtraj"]).reshape(N + 1, nx)
                e8 = (np.abs(g8["xtraj"][0] - x8) / np.maximum(np.abs(x8).max(axis=1, keepdims=True), 1.0)).max()
                if form == 1:                           # like for like: every integer, 1e-8
                    assert g8["qp_iter_total"][0] == r8["qp_iter_total"] and e8 < 1e-8, (c["config"], form, e8)
                else:                                   # the other form of the recursion: the same iterate (an interior-point count may differ by one where a residual sits at the tolerance)
                    assert abs(int(g8["qp_iter_total"][0]) - r8["qp_iter_total"]) <= 1 and e8 < 1e-6, (c["config"], form, e8)
