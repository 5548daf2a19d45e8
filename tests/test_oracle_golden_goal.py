"""CPU oracle on the reference's model WITHOUT a spline state: SecondOrderUnicycleModel (solver_model.py:170-191) with the goal-tracking
stack (MPC base + goal_module.py:22-36 + ellipsoids) vs golden vectors made by executing the reference's own python (tests/golden/make_golden_goal.py).
The oracle -- like the HIP kernels -- keeps its 5-state arrays for this model with the fifth slot INERT: s' = 0, nothing reads it.  Checked here: on
the six real variables every stage function (cost, rows, ERK4 x 3 dynamics; values, gradients, Hessians) equals the reference's, and the padding
slot contributes exactly nothing (zero gradient / Hessian rows, x_next[4] = z[6]); a solve keeps the slot at 0 and is feasible."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "stage_functions_goal.json")) as fh:
    GOLD = json.load(fh)
CASES = GOLD["cases"]
IDS = [c["name"] for c in CASES]


def pb_for(case):
    pb = O.problem(N=case["N"], S=5, n_lin=0, M=case["M"], goal_stack=1)
    assert pb.npar == case["npar"] == 37 and pb.nh == case["nh"] and pb.model == 1 and pb.cost_model == 2
    return pb


def test_model_bounds_are_the_reference_models():
    pb = pb_for(CASES[0])
    assert list(pb.lb)[:6] == GOLD["model"]["lower_bound"] and list(pb.ub)[:6] == GOLD["model"]["upper_bound"]
    pm = CASES[0]["parameter_map"]
    assert [pm[n] for n in ("acceleration", "angular_velocity", "velocity", "reference_velocity", "goal_weight", "goal_x", "goal_y",
                            "ego_disc_radius", "ego_disc_0_offset", "ellipsoid_obst_0_x", "ellipsoid_obst_3_r")] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 36]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_stage_functions_equal_the_references_on_the_real_variables(case):
    pb = pb_for(case)
    for pad in (0.0, 3.25):                                    # whatever the padding slot holds, the real variables' functions do not see it
        z = np.array(case["z"] + [pad])
        v, g, H = O.stage_cost(pb, z, case["p"])
        np.testing.assert_allclose(v, case["cost"], rtol=1e-12)
        np.testing.assert_allclose(g[:6], case["cost_grad"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(H[:6, :6], case["cost_hess"], rtol=1e-9, atol=1e-10)
        assert g[6] == 0.0 and not H[6].any() and not H[:, 6].any()
        h, J, HH = O.stage_constraints(pb, z, case["p"])
        np.testing.assert_allclose(h, case["h"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(J[:, :6], case["h_jac"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(HH[:, :6, :6], case["h_hess"], rtol=1e-9, atol=1e-10)
        assert not J[:, 6].any() and not HH[:, 6].any() and not HH[:, :, 6].any()
        xn, Jd, Hd = O.discrete_dynamics(pb, z)
        np.testing.assert_allclose(xn[:4], case["x_next"], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(Jd[:4, :6], case["x_next_jac"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(Hd[:4, :6, :6], case["x_next_hess"], rtol=1e-10, atol=1e-12)
        assert xn[4] == pad and np.array_equal(Jd[4], np.eye(7)[6]) and not Hd[4].any()      # the padding slot stands still and couples with nothing
        assert not Jd[:4, 6].any() and not Hd[:4, 6].any() and not Hd[:4, :, 6].any()


def test_solve_keeps_the_padding_slot_inert_and_reaches_for_the_goal():
    from mpc_planner_amd import modules as md
    pb = pb_for(CASES[0]); N, B = 20, 8
    params = np.zeros((B, N, pb.npar))
    for i, v in enumerate([0.34, 0.85, 0.55, 2.0, 4.0, 9.0, 0.5, 0.325, 0.0]):
        params[:, :, i] = v
    rng = np.random.default_rng(4)
    for b in range(B):
        for j in range(4):
            params[b, 1:, 9 + 7 * j:16 + 7 * j] = [2.5 + 1.8 * j, (-1) ** (j + b) * rng.uniform(0.9, 1.6), 0.0, 0.0, 0.0, 1.0, 0.4]
            params[b, 0, 9 + 7 * j:16 + 7 * j] = [50.0, 50.0, 0.0, 0.0, 0.0, 1.0, 0.1]             # stage 0: far-away dummies
    xinit = np.zeros((B, 5)); xinit[:, 3] = rng.uniform(0.5, 1.5, B)
    x0 = np.stack([md.initialize_with_forward_propagation(xinit[b], N, 0.2) for b in range(B)]); x0[:, :, 6] = 0.0
    xt, ut, info = O.solve_batch(pb, xinit, x0.reshape(B, -1), params.reshape(B, -1))
    assert (info["exit_code"] == 1).all() and not xt[:, :, 4].any()
    for b in range(B):
        for k in range(1, N):
            h, _, _ = O.stage_constraints(pb, np.concatenate([ut[b, k], xt[b, k], [0.0]]), params[b, k])
            assert h.min() > 1.0 - 1e-5                                          # every ellipsoid row h >= 1
        assert np.hypot(9.0 - xt[b, N, 0], 0.5 - xt[b, N, 1]) < np.hypot(9.0 - x0[b, N, 2], 0.5 - x0[b, N, 3])
