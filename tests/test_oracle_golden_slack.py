"""CPU oracle (slack build, oracle/liboracle_tmpc_slack.so) vs golden vectors made by executing the reference's own
python modules for the slack-model configurations (tests/golden/make_golden_slack.py): BASELINE config 3
(rosnavigation T-MPC: guidance + ellipsoids + decomp), config 5 (SH-MPC: 24 scenario halfspaces) and the
rosnavigation safe-horizon variant (scenario + decomp)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "stage_functions_slack.json")) as fh:
    CASES = json.load(fh)["cases"]
IDS = [c["name"] for c in CASES]


def pb_for(case):
    pb = O.problem(N=case["N"], S=case["S"], n_lin=case["n_lin"], M=case["M"], n_slk=case["n_scen"] + case["n_dec"], slack=1)
    assert pb.npar == case["npar"]            # 172 / 127 / 163 (cfg 3 = 172, cfg 5 = 127)
    assert pb.nh == case["nh"] and pb.nxe == 6 and pb.nve == 8
    return pb


def close(a, b, rtol=1e-11, atol=1e-12):
    np.testing.assert_allclose(np.asarray(a, float), np.asarray(b, float), rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_parameter_map_matches_reference(case):
    pb = pb_for(case); L = O.lib(1); pm = case["parameter_map"]
    names = ["acceleration", "angular_velocity", "velocity", "reference_velocity", "contour", "lag",
             "terminal_angle", "terminal_contouring", "slack"]
    for i, n in enumerate(names):
        assert L.orc_idx_weight(C.byref(pb), i) == pm[n]
    sp = ["spline_x{}_a", "spline_x{}_b", "spline_x{}_c", "spline_x{}_d", "spline_y{}_a", "spline_y{}_b",
          "spline_y{}_c", "spline_y{}_d", "spline{}_start"]
    for s in range(case["S"]):
        for w, n in enumerate(sp):
            assert L.orc_idx_spline(C.byref(pb), s, w) == pm[n.format(s)]
    for j in range(case["n_lin"]):
        for w, n in enumerate(["a1", "a2", "b"]):
            assert L.orc_idx_lin(C.byref(pb), j, w) == pm[f"lin_constraint_{j}_{n}"]
    if case["M"]:
        assert L.orc_idx_disc_radius(C.byref(pb)) == pm["ego_disc_radius"]
    assert L.orc_idx_disc_offset(C.byref(pb)) == pm["ego_disc_0_offset"]
    for j in range(case["M"]):
        for w, n in enumerate(["x", "y", "psi", "major", "minor", "chi", "r"]):
            assert L.orc_idx_ellipsoid(C.byref(pb), j, w) == pm[f"ellipsoid_obst_{j}_{n}"]
    # scenario rows come before decomp rows (module order of generate_rosnavigation_solver.py:82-83)
    rows = [f"disc_0_scenario_constraint_{j}" for j in range(case["n_scen"])] + [f"disc_0_decomp_{j}" for j in range(case["n_dec"])]
    for j, r in enumerate(rows):
        for w, n in enumerate(["a1", "a2", "b"]):
            assert L.orc_idx_slk(C.byref(pb), j, w) == pm[f"{r}_{n}"]
    # bounds of the slack model (solver_model.py:285-286)
    close(list(pb.lb) + [pb.lb_slack], case["lower_bound"]); close(list(pb.ub) + [pb.ub_slack], case["upper_bound"])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_stage_cost(case):
    pb = pb_for(case)
    v, g, H = O.stage_cost(pb, case["z"], case["p"])
    close(v, case["cost"]); close(g, case["cost_grad"], atol=1e-10); close(H, case["cost_hess"], atol=1e-9)
    # the slack row/column of the cost Hessian is diagonal (what the QP presolve of sqp_rti.c U9 relies on)
    assert np.all(H[7, :7] == 0.0) and np.all(H[:7, 7] == 0.0) and H[7, 7] == 2.0 * 10000.0


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_stage_constraints(case):
    pb = pb_for(case)
    h, J, H = O.stage_constraints(pb, case["z"], case["p"])
    close(h, case["h"]); close(J, case["h_jac"], atol=1e-11); close(H, case["h_hess"], atol=1e-11)
    assert np.all(H[:, 7, :] == 0.0) and np.all(H[:, :, 7] == 0.0)       # slack enters the rows linearly
    lh = np.zeros(case["nh"]); uh = np.zeros(case["nh"])
    O.lib(1).orc_constraint_bounds(C.byref(pb), O.dptr(lh), O.dptr(uh))
    close(lh, case["lh"]); close(uh, case["uh"])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_dynamics(case):
    pb = pb_for(case)
    close(O.continuous_dynamics(case["z"], slack=1), case["f_cont"])
    xn, J, H = O.discrete_dynamics(pb, case["z"])
    close(xn, case["x_next"]); close(J, case["x_next_jac"], atol=1e-13); close(H, case["x_next_hess"], atol=1e-13)
    assert xn[5] == case["z"][7] and J[5, 7] == 1.0 and np.count_nonzero(J[5]) == 1    # slack' = 0
