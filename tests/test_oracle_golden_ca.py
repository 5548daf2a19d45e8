"""CPU oracle, curvature-aware contouring cost (orc_problem.cost_model = 1) vs golden vectors made by executing the reference's own
curvature_aware_contouring.py (tests/golden/make_golden_ca.py): BASELINE configs[2] "Jackal CA-MPC + decomp_util static constraints"
(slack model, npar 172) and the same stack on the plain unicycle (npar 135)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "stage_functions_ca.json")) as fh:
    CASES = json.load(fh)["cases"]
IDS = [c["name"] for c in CASES]


def pb_for(case, **kw):
    pb = O.problem(N=case["N"], S=case["S"], n_lin=case["n_lin"], M=case["M"], n_slk=case["n_dec"], slack=case["slack"], cost_model=1, **kw)
    assert pb.npar == case["npar"] and pb.nh == case["nh"]
    return pb


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_parameter_map_is_the_mpcc_stacks(case):
    """CurvatureAwareContouringObjective.define_parameters adds contour, lag, [velocity, reference_velocity unless MPCBase did,]
    terminal_*, then the spline rows: with MPCBase weighing v the map equals the ContouringModule stack's."""
    pb = pb_for(case); L = O.lib(case["slack"]); pm = case["parameter_map"]
    names = ["acceleration", "angular_velocity", "velocity", "reference_velocity", "contour", "lag", "terminal_angle", "terminal_contouring"]
    for i, n in enumerate(names):
        assert L.orc_idx_weight(C.byref(pb), i) == pm[n]
    assert L.orc_idx_spline(C.byref(pb), 0, 0) == pm["spline_x0_a"] and L.orc_idx_spline(C.byref(pb), 4, 8) == pm["spline4_start"]
    assert L.orc_idx_lin(C.byref(pb), 0, 0) == pm["lin_constraint_0_a1"] and L.orc_idx_ellipsoid(C.byref(pb), 7, 6) == pm["ellipsoid_obst_7_r"]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_curvature_aware_stage_cost(case):
    pb = pb_for(case)
    v, g, H = O.stage_cost(pb, case["z"], case["p"])
    np.testing.assert_allclose(v, case["cost"], rtol=1e-12)
    np.testing.assert_allclose(g, case["cost_grad"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(H, case["cost_hess"], rtol=1e-9, atol=1e-10)
    # what distinguishes it structurally from the MPCC cost: psi and v couple with (x, y, s) through s_dot -> MIRROR sees a full 7 x 7 block
    assert abs(H[4, 6]) > 0 and abs(H[5, 2]) > 0
    # rows are the MPCC stack's (same modules)
    h, _, _ = O.stage_constraints(pb, case["z"], case["p"])
    np.testing.assert_allclose(h, case["h"], rtol=1e-11, atol=1e-12)


def test_cost_model_switch_changes_the_cost_only():
    case = CASES[0]
    a = O.stage_cost(pb_for(case), case["z"], case["p"])[0]
    pb0 = pb_for(case); pb0.cost_model = 0
    b = O.stage_cost(pb0, case["z"], case["p"])[0]
    assert a != b


def test_oracle_solves_the_ca_configuration():
    """configs[2] as named: the CA stack solves on the cfg-3 scenes (N = 30, slack model, 8 + 8 + 12 rows) -- success, feasible rows,
    and a different optimum than the MPCC stack's on the same parameters."""
    from mpc_planner_amd import scenes
    sc = scenes.make_scene(1, N=30, M=8, B=8, slack=True, n_decomp=12)
    kw = dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1)
    xt, ut, info = O.solve_batch(O.problem(cost_model=1, **kw), sc["xinit"], sc["x0"].reshape(8, -1), sc["params"].reshape(8, -1))
    xt0, _, info0 = O.solve_batch(O.problem(**kw), sc["xinit"], sc["x0"].reshape(8, -1), sc["params"].reshape(8, -1))
    ok = info["exit_code"] == 1
    assert ok.mean() >= 0.75 and (info["res_eq"][ok] < 1e-2).all() and (info["sqp_iter"][ok] == 10).all()
    both = ok & (info0["exit_code"] == 1)
    assert both.any() and np.abs(xt[both] - xt0[both]).max() > 1e-3
