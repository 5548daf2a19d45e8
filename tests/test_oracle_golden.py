"""CPU oracle vs golden vectors produced by executing the reference's own python modules
(tests/golden/make_golden.py) and vs the hand-checkable anchors of the reference's tests."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "stage_functions.json")) as fh:
    GOLD = json.load(fh)
CASES = GOLD["cases"]


def pb_for(case):
    n_lin = case["M"] if case["uses_lin_rows"] else 0
    pb = O.problem(N=20, S=case["S"], n_lin=n_lin, M=case["M"])
    assert pb.npar == case["npar"]            # 83 / 135 / 175 (reference Parameters order)
    assert n_lin + case["M"] == case["nh"]
    return pb


def close(a, b, rtol=1e-11, atol=1e-12):
    np.testing.assert_allclose(np.asarray(a, float), np.asarray(b, float), rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_parameter_map_matches_reference(case):
    pb = pb_for(case); L = O.lib(); pm = case["parameter_map"]
    import ctypes as C
    names = ["acceleration", "angular_velocity", "velocity", "reference_velocity", "contour", "lag",
             "terminal_angle", "terminal_contouring"]
    for i, n in enumerate(names):
        assert L.orc_idx_weight(C.byref(pb), i) == pm[n]
    sp = ["spline_x{}_a", "spline_x{}_b", "spline_x{}_c", "spline_x{}_d", "spline_y{}_a", "spline_y{}_b",
          "spline_y{}_c", "spline_y{}_d", "spline{}_start"]
    for s in range(case["S"]):
        for w, n in enumerate(sp):
            assert L.orc_idx_spline(C.byref(pb), s, w) == pm[n.format(s)]
    if case["uses_lin_rows"]:
        for j in range(case["M"]):
            for w, n in enumerate(["a1", "a2", "b"]):
                assert L.orc_idx_lin(C.byref(pb), j, w) == pm[f"lin_constraint_{j}_{n}"]
    assert L.orc_idx_disc_radius(C.byref(pb)) == pm["ego_disc_radius"]
    assert L.orc_idx_disc_offset(C.byref(pb)) == pm["ego_disc_0_offset"]
    for j in range(case["M"]):
        for w, n in enumerate(["x", "y", "psi", "major", "minor", "chi", "r"]):
            assert L.orc_idx_ellipsoid(C.byref(pb), j, w) == pm[f"ellipsoid_obst_{j}_{n}"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_stage_cost(case):
    pb = pb_for(case)
    v, g, H = O.stage_cost(pb, case["z"], case["p"])
    close(v, case["cost"]); close(g, case["cost_grad"], atol=1e-11); close(H, case["cost_hess"], atol=1e-10)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_stage_constraints(case):
    pb = pb_for(case)
    h, J, H = O.stage_constraints(pb, case["z"], case["p"])
    close(h, case["h"]); close(J, case["h_jac"], atol=1e-11); close(H, case["h_hess"], atol=1e-11)
    import ctypes as C
    lh = np.zeros(case["nh"]); uh = np.zeros(case["nh"])
    O.lib().orc_constraint_bounds(C.byref(pb), O.dptr(lh), O.dptr(uh))
    close(lh, case["lh"]); close(uh, case["uh"])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_dynamics(case):
    pb = pb_for(case)
    close(O.continuous_dynamics(case["z"]), case["f_cont"])
    xn, J, H = O.discrete_dynamics(pb, case["z"])
    close(xn, case["x_next"]); close(J, case["x_next_jac"], atol=1e-13); close(H, case["x_next_hess"], atol=1e-13)


def test_reference_test_anchors():
    """Inputs of solver_generator/test/test_control_modules.py:56-59 and :89-95; the exact values
    (25.0, 125.0) follow from the reference's module code."""
    anc = GOLD["reference_test_anchors"]
    assert anc["test_module_manager_objective"]["npar"] == 10 * 9 + 2 + 4 + 10 * 4
    assert abs(anc["test_module_manager_objective"]["objective"] - 25.0) < 1e-12
    assert anc["test_module_manager_constraints"]["npar"] == 7 + 2
    # contouring-only objective with all parameters = 1, z[3] = y = 5 (10 segments): the oracle's cost
    # with zero MPCBase weights must reproduce 25.0.  Oracle layout needs the 4 base weights in front.
    S = 10
    pb = O.problem(N=20, S=S, n_lin=0, M=1)
    p = np.ones(pb.npar); p[0] = p[1] = p[2] = 0.0      # w_a = w_w = w_v = 0 -> contouring terms only
    z = np.zeros(7); z[3] = 5.0
    v, _, _ = O.stage_cost(pb, z, p)
    assert abs(v - 25.0) < 1e-12
    # ellipsoid: p[2]=obst_x=5, p[3]=obst_y=10, r=1 -> (25+100)/1 = 125
    pb = O.problem(N=20, S=1, n_lin=0, M=1)
    p = np.zeros(pb.npar)
    import ctypes as C
    L = O.lib()
    p[L.orc_idx_ellipsoid(C.byref(pb), 0, 0)] = 5.0; p[L.orc_idx_ellipsoid(C.byref(pb), 0, 1)] = 10.0
    p[L.orc_idx_ellipsoid(C.byref(pb), 0, 6)] = 1.0
    h, _, _ = O.stage_constraints(pb, np.zeros(7), p)
    assert abs(h[0] - 125.0) < 1e-12 and anc["test_module_manager_constraints"]["constraint"] == [125.0]


def test_survey_appendix_e_known_answers():
    """SURVEY Appendix E vectors (reference module code executed numerically during the survey)."""
    pb = O.problem(N=20, S=5, n_lin=8, M=8)
    import ctypes as C
    L = O.lib(); p = np.zeros(pb.npar)
    p[:8] = [0.34, 0.85, 0.55, 2.0, 0.05, 0.75, 100.0, 10.0]
    for i in range(5):
        vals = [0, 0, 1, 6 * i, 0.002 * (i + 1), -0.01 * (i + 1), 0.05, 0.1 * i, 6 * i]
        for w, v in enumerate(vals):
            p[L.orc_idx_spline(C.byref(pb), i, w)] = v
    p[L.orc_idx_disc_radius(C.byref(pb))] = 0.325
    g = np.array([2.5, 0.2])
    for j in range(8):
        o = np.array([3 + 1.5 * j, (-1) ** j * (1 + 0.25 * j)])
        vals = [o[0], o[1], 0.3 * j, 0.1 * j, 0.05 * j, 5.991464547107979 if j % 2 else 1.0, 0.4]
        for w, v in enumerate(vals):
            p[L.orc_idx_ellipsoid(C.byref(pb), j, w)] = v
        a = (o - g) / np.linalg.norm(o - g)
        for w, v in enumerate([a[0], a[1], a @ o - (1e-3 + 0.325)]):
            p[L.orc_idx_lin(C.byref(pb), j, w)] = v
    z = [0.3, -0.2, 2.4, 0.35, 0.15, 1.7, 2.6]
    v, _, _ = O.stage_cost(pb, z, p)
    assert abs(v - 0.14458957750089863) < 1e-14
    h, _, _ = O.stage_constraints(pb, z, p)
    close(h[:8], [-0.54319826160521534, -2.3133300454657038, -3.4491454882793278, -5.1884633116770269,
                  -6.474968631307398, -8.1802901159513031, -9.5103525311904296, -11.198141804753913], rtol=1e-13)
    close(h[8:], [1.4887039239001187, 7.6827278627892692, 19.364362355982522, 16.951938637165501,
                  53.816105685106415, 37.51931817790674, 85.134014078548105, 51.104828894149065], rtol=1e-13)
    close(O.continuous_dynamics(z), [1.6809108324912718, 0.25404482520511867, -0.2, 0.3, 1.7], rtol=1e-14)
    z2 = [-0.5, 0.4, 6.1, -0.4, -0.3, 0.9, 5.97]
    v2, _, _ = O.stage_cost(pb, z2, p)
    assert abs(v2 - 0.91082299672420519) < 1e-13


def test_mirror_matches_eigh():
    rng = np.random.default_rng(0)
    for n in (5, 7):
        for _ in range(20):
            A = rng.normal(size=(n, n)); A = A + A.T
            A[0, 0] = 1e-6  # small entries too
            e, V = np.linalg.eigh(A)
            e2 = np.where(np.abs(e) <= 1e-4, 1e-4, np.abs(e))
            close(O.mirror(A), (V * e2) @ V.T, rtol=1e-10, atol=1e-12)
    close(O.mirror(np.zeros((5, 5))), 1e-4 * np.eye(5))


def test_gaussian_chance_rows_match_reference_golden():
    """mpc_planner_jackal's default stack (guidance + GaussianConstraintModule, generate_jackal_solver.py:53-73): oracle rows,
    parameter map and bounds against the reference's own scripts (tests/golden/make_golden_gaussian.py)."""
    import ctypes as C
    with open(os.path.join(HERE, "golden", "stage_functions_gaussian.json")) as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        pb = O.problem(N=case["N"], S=case["S"], n_lin=case["M"], M=0, n_gauss=case["M"])
        assert pb.npar == case["npar"] == 82 and pb.nh == case["nh"] == 10
        pm = case["parameter_map"]; L = O.lib()
        assert L.orc_idx_disc_radius(C.byref(pb)) == pm["ego_disc_radius"]
        for j in range(case["M"]):
            for w, n in enumerate(["x", "y", "major", "minor", "risk", "r"]):
                assert L.orc_idx_gaussian(C.byref(pb), j, w) == pm[f"gaussian_obst_{j}_{n}"]
        v, g, H = O.stage_cost(pb, case["z"], case["p"])
        close(v, case["cost"]); close(H, case["cost_hess"], atol=1e-10)
        h, J, Hh = O.stage_constraints(pb, case["z"], case["p"])
        close(h, case["h"], rtol=1e-10); close(J, case["h_jac"], rtol=1e-9, atol=1e-11); close(Hh, case["h_hess"], rtol=1e-8, atol=1e-10)
        lh = np.zeros(10); uh = np.zeros(10)
        L.orc_constraint_bounds(C.byref(pb), O.dptr(lh), O.dptr(uh))
        close(lh, case["lh"]); close(uh, case["uh"])
