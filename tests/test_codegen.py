"""Solver generation: module stacks written against the reference's plugin protocol -> emitted stage
functions.  The emitted C++ is compiled for the host and compared with the golden vectors produced by executing the
reference's own module scripts (tests/golden/*.json): parameter maps, values, gradients, Hessians."""
import json
import os

import numpy as np
import pytest

from mpc_planner_amd.codegen import emit, stacks
from mpc_planner_amd.codegen.hostlib import HostStageFunctions

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(fname, config):
    with open(os.path.join(HERE, "golden", fname)) as fh:
        return [c for c in json.load(fh)["cases"] if config is None or c["config"] == config]


def _check(gen, cases, slack_model):
    hs = HostStageFunctions(gen["header"])
    assert hs.npar == cases[0]["npar"] and dict(gen["params"]._params) == cases[0]["parameter_map"]
    for c in cases:
        z = np.array(c["z"]); slack = z[7] if slack_model else 0.0
        v, g, H = hs.cost(z, c["p"], slack)
        np.testing.assert_allclose(v, c["cost"], rtol=1e-12)
        np.testing.assert_allclose(g, np.array(c["cost_grad"])[:7], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(H, np.array(c["cost_hess"])[:7, :7], rtol=1e-9, atol=1e-10)
        h, D, Hr = hs.rows(z, c["p"], slack)
        hh, J, HH = np.array(c["h"]), np.array(c["h_jac"]), np.array(c["h_hess"])
        lh, uh = np.array(c["lh"]), np.array(c["uh"])
        assert hs.nh == int((np.abs(lh) < 1e14).sum() + (np.abs(uh) < 1e14).sum())
        for k in range(hs.nh):                      # emitted row k is  sign * (h_src - bound) <= 0
            r, s = hs.row_src[k], hs.row_sign[k]
            assert hs.row_bound[k] == (uh[r] if s > 0 else lh[r])
            np.testing.assert_allclose(h[k], s * (hh[r] - hs.row_bound[k]), rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(D[k], s * J[r][[2, 3, 4]], rtol=1e-10, atol=1e-11)
            np.testing.assert_allclose(Hr[k], s * HH[r][np.ix_([2, 3, 4], [2, 3, 4])], rtol=1e-10, atol=1e-11)
            assert np.all(J[r][[0, 1, 5, 6]] == 0.0)      # what the generator verified symbolically


def test_generated_tmpc_stack_matches_reference_golden():
    """BASELINE cfg 2 stack (MPC base + contouring + guidance halfspaces + ellipsoids), incl. the point 3 cm before a
    spline knot and points several segments away from one (the glue sigmoids' derivatives must not overflow)."""
    st = stacks.settings(N=20, max_obstacles=8)
    model, mm = stacks.tmpc(st)
    _check(emit.generate(mm, model, st, "cfg2", method="jets"), _golden("stage_functions.json", "cfg2_tmpc_M8"), False)


def test_generated_safe_horizon_stack_matches_reference_golden():
    """BASELINE cfg 5 stack (slack model, 24 scenario halfspaces): the slack enters as the per-trajectory constant."""
    st = stacks.settings(N=20, max_obstacles=8)
    model, mm = stacks.safe_horizon(st)
    gen = emit.generate(mm, model, st, "cfg5", method="jets")
    assert gen["slack"] == 1 and gen["nh"] == 24
    _check(gen, _golden("stage_functions_slack.json", "cfg5_safe_horizon"), True)


def test_generated_jackal_gaussian_tmpc_stack_matches_reference_golden():
    """mpc_planner_jackal's default stack (guidance halfspaces + Gaussian chance constraints as the submodule)."""
    st = stacks.settings(N=30, max_obstacles=5, num_segments=3)
    model, mm = stacks.jackal_tmpc(st)
    gen = emit.generate(mm, model, st, "jackal_tmpc", method="jets")
    assert gen["npar"] == 82 and gen["nh"] == 10
    cases = _golden("stage_functions_gaussian.json", None)
    _check(gen, cases, False)


def test_generated_goal_gaussian_stack_against_finite_differences():
    """A stack with modules the hand-written kernels do not have (GoalModule, GaussianConstraintModule): emitted
    derivatives against central differences of the emitted values; erf / log / sqrt paths of the chance constraint."""
    st = stacks.settings(N=20, max_obstacles=3)
    model, mm = stacks.goal_gaussian(st)
    gen = emit.generate(mm, model, st, "goal_gaussian", method="jets")
    pm = gen["params"]
    hs = HostStageFunctions(gen["header"])
    assert hs.nh == 3 and hs.npar == pm.length() == 4 + 3 + 2 + 6 * 3
    rng = np.random.default_rng(1)
    p = np.zeros(hs.npar)
    for n, v in dict(acceleration=0.3, angular_velocity=0.8, velocity=0.5, reference_velocity=2.0, goal_weight=1.5, goal_x=7.0,
                     goal_y=-1.0, ego_disc_radius=0.325, ego_disc_0_offset=0.1).items():
        p[pm.index(n)] = v
    for j in range(3):
        for f, v in dict(x=3.0 + 2 * j, y=(-1) ** j * 1.5, major=0.4 + 0.1 * j, minor=0.2, risk=0.05, r=0.4).items():
            p[pm.index(f"gaussian_obst_{j}_{f}")] = v
    z = np.array([0.4, -0.2, 1.0, 0.3, 0.2, 1.5, 2.0])
    v, g, H = hs.cost(z, p)
    assert abs(v - (0.3 * 0.16 + 0.8 * 0.04 + 0.5 * 0.25 + 1.5 * (36.0 + 1.69) / (49.0 + 1.0 + 0.01))) < 1e-12
    eps = 1e-6
    for i in range(7):
        e = np.zeros(7); e[i] = eps
        vp, gp, _ = hs.cost(z + e, p); vm, gm, _ = hs.cost(z - e, p)
        assert abs((vp - vm) / (2 * eps) - g[i]) < 1e-7
        np.testing.assert_allclose((gp - gm) / (2 * eps), H[:, i], atol=1e-6)
    h, D, Hr = hs.rows(z, p)
    assert np.all(hs.row_sign == -1) and np.all(hs.row_bound == 0.0)         # chance constraints are >= 0 rows
    for a, i in enumerate((2, 3, 4)):
        e = np.zeros(7); e[i] = eps
        hp, Dp, _ = hs.rows(z + e, p); hm, Dm, _ = hs.rows(z - e, p)
        np.testing.assert_allclose((hp - hm) / (2 * eps), D[:, a], atol=1e-7)
        np.testing.assert_allclose((Dp - Dm) / (2 * eps), Hr[:, :, a], atol=1e-6)
    # the row itself: distance along the line of sight minus radii minus the Gaussian margin (erfinv(1 - 2 risk) sqrt(2 a^T Sigma a))
    from scipy.special import erfinv
    c, s = np.cos(z[4]), np.sin(z[4])
    for j in range(3):
        o = np.array([3.0 + 2 * j, (-1) ** j * 1.5]); d = np.array([z[2] + 0.1 * c, z[3] + 0.1 * s]) - o
        a = d / np.linalg.norm(d)
        want = np.linalg.norm(d) - (0.325 + 0.4) - erfinv(1 - 2 * 0.05) * np.sqrt(2 * (a[0] ** 2 * (0.4 + 0.1 * j) ** 2 + a[1] ** 2 * 0.04))
        assert abs(-h[j] - want) < 1e-9


def test_jet_and_symbolic_differentiation_agree():
    """The two differentiation back ends of emit.py (sparse forward-mode jets at code level -- the default -- and sympy
    derivatives + CSE) are independent implementations: same values, gradients and Hessians on a stack with sqrt, log, erf,
    exp, sin / cos and rational functions."""
    st = stacks.settings(N=20, max_obstacles=2)
    model, mm = stacks.goal_gaussian(st)
    a = HostStageFunctions(emit.generate(mm, model, st, "a", method="jets")["header"])
    model, mm = stacks.goal_gaussian(st)
    b = HostStageFunctions(emit.generate(mm, model, st, "b", method="symbolic")["header"])
    rng = np.random.default_rng(2)
    for _ in range(5):
        p = rng.uniform(0.2, 1.5, a.npar); z = rng.uniform(-1.0, 2.0, 7)
        for j in range(2):
            p[a.npar - 6 * (2 - j) + 4] = rng.uniform(0.01, 0.3)       # risk in (0, 0.5)
        for fa, fb in zip(a.cost(z, p), b.cost(z, p)):
            np.testing.assert_allclose(fa, fb, rtol=1e-11, atol=1e-12)
        for fa, fb in zip(a.rows(z, p), b.rows(z, p)):
            np.testing.assert_allclose(fa, fb, rtol=1e-10, atol=1e-11)


def test_generator_refuses_rows_outside_the_kernel_structure():
    from mpc_planner_amd.codegen import plugin as P, library as L

    class VelocityRow:
        nh = 1
        def define_parameters(self, params): params.add("vmax")
        def get_lower_bound(self): return [-np.inf]
        def get_upper_bound(self): return [0.0]
        def get_constraints(self, model, params, settings, stage_idx): return [model.get("v") - params.get("vmax")]

    class M(P.ConstraintModule):
        def __init__(self):
            super().__init__(); self.constraints.append(VelocityRow())
    st = stacks.settings()
    mm = P.ModuleManager(); mm.add_module(L.GoalModule(st)); mm.add_module(M())
    with pytest.raises(emit.UnsupportedStack, match="depends on `v`"):
        emit.generate(mm, P.UnicycleContouringModel(), st, "bad")


def test_cpp_glue_files_for_the_tmpc_stack(tmp_path):
    """modules.h / definitions.h / modules.cmake (generate_cpp_files.py:11-95): factory in stack order, the guidance module's
    extra headers, WEIGHT_PARAMS and GUIDANCE_CONSTRAINTS_TYPE, dependency / source lists without duplicates."""
    from mpc_planner_amd.codegen import cpp_glue, plugin as P
    st = stacks.settings()
    _, mm = stacks.rosnav_tmpc(st)
    files = cpp_glue.write_module_glue(str(tmp_path), mm)
    assert [os.path.basename(f) for f in files] == ["definitions.h", "modules.h", "modules.cmake"]
    mh = open(tmp_path / "include" / "mpc_planner_modules" / "modules.h").read()
    order = [mh.index(f"std::make_shared<{n}>(solver)") for n in ("MPCBaseModule", "Contouring", "GuidanceConstraints", "DecompConstraints")]
    assert order == sorted(order)
    for inc in ("mpc_base.h", "contouring.h", "guidance_constraints.h", "linearized_constraints.h", "ellipsoid_constraints.h", "decomp_constraints.h"):
        assert f"#include <mpc_planner_modules/{inc}>" in mh
    assert "inline void initializeModules(std::vector<std::shared_ptr<ControllerModule>> &modules, std::shared_ptr<Solver> solver)" in mh
    dh = open(tmp_path / "include" / "mpc_planner_modules" / "definitions.h").read()
    assert '#define WEIGHT_PARAMS {"acceleration", "angular_velocity", "slack", "velocity", "reference_velocity"}' in dh
    assert "#define GUIDANCE_CONSTRAINTS_TYPE EllipsoidConstraints" in dh
    cm = open(tmp_path / "modules.cmake").read()
    assert cm.count("guidance_planner") == 2 and cm.count("decomp_util") == 2          # find_package + dependency list
    for src in ("mpc_base", "contouring", "guidance_constraints", "linearized_constraints", "ellipsoid_constraints", "decomp_constraints"):
        assert f"\tsrc/{src}.cpp\n" in cm
    # a python-only module (no C++ counterpart) is left out of the factory
    class PyOnly(P.ObjectiveModule):
        pass
    mm.add_module(PyOnly())
    assert "UNDEFINED" not in cpp_glue.modules_header(mm)


def test_reference_generation_test_configurations():
    """The assertions of the reference's own generator tests, on this generator: parameter count of contouring + path
    reference velocity (test_control_modules.py:27-53) and the stack of test_acados.py:30-77 (12 ellipsoid rows, nx = 5,
    nu = 2); with a dynamic velocity reference the cost tracks the path's velocity spline (contouring.py:60-81)."""
    from mpc_planner_amd.codegen import plugin as P, library as L
    st = {"N": 20, "contouring": {"num_segments": 10, "dynamic_velocity_reference": False}}
    mm = P.ModuleManager(); mm.add_module(L.ContouringModule(st)); mm.add_module(L.PathReferenceVelocityModule(st))
    pm = P.define_parameters(mm, P.Parameters(), st)
    assert pm.length() == 10 * 9 + 2 + 4 + 10 * 4
    assert [pm.index(n) for n in ("contour", "lag", "velocity", "reference_velocity", "terminal_angle", "terminal_contouring")] == list(range(6))

    st = stacks.settings(N=20, max_obstacles=12, num_segments=8)
    model, mm = stacks.contouring_path_velocity_ellipsoids(st)
    gen = emit.generate(mm, model, st, "test_solver", method="jets")
    assert gen["nh"] == 12 and model.nx == 5 and model.nu == 2
    assert gen["npar"] == 2 + 6 + 8 * 9 + 8 * 4 + 2 + 12 * 7

    # dynamic velocity reference: value by hand on a straight path with v_ref(s) = 1.5 + 0.25 s, derivatives by differences
    st = stacks.settings(N=20, max_obstacles=1, num_segments=2); st["contouring"]["dynamic_velocity_reference"] = True
    model, mm = stacks.contouring_path_velocity_ellipsoids(st)
    gen = emit.generate(mm, model, st, "dynvel", method="jets")
    pm, hs = gen["params"], HostStageFunctions(gen["header"])
    p = np.zeros(hs.npar)
    for n, v in dict(acceleration=0.3, angular_velocity=0.8, contour=0.05, lag=0.75, velocity=0.55, ego_disc_radius=0.3,
                     ellipsoid_obst_0_x=50.0, ellipsoid_obst_0_y=50.0, ellipsoid_obst_0_r=0.1, ellipsoid_obst_0_chi=1.0).items():
        p[pm.index(n)] = v
    for i, start in enumerate((0.0, 6.0)):                  # x(s) = s, y(s) = 0, v_ref(s) = 1.5 + 0.25 s on both segments
        p[pm.index(f"spline_x{i}_c")] = 1.0; p[pm.index(f"spline_x{i}_d")] = start
        p[pm.index(f"spline_v{i}_c")] = 0.25; p[pm.index(f"spline_v{i}_d")] = 1.5 + 0.25 * start
        p[pm.index(f"spline{i}_start")] = start
    z = np.array([0.4, -0.2, 2.3, 0.6, 0.1, 1.2, 2.0])      # a, w, x, y, psi, v, s
    v, g, H = hs.cost(z, p)
    want = 0.3 * 0.16 + 0.8 * 0.04 + 0.75 * 0.3 ** 2 + 0.05 * 0.6 ** 2 + 0.55 * (1.2 - 2.0) ** 2
    assert abs(v - want) < 1e-9
    eps = 1e-6
    for i in range(7):
        e = np.zeros(7); e[i] = eps
        vp, gp, _ = hs.cost(z + e, p); vm, gm, _ = hs.cost(z - e, p)
        assert abs((vp - vm) / (2 * eps) - g[i]) < 1e-7
        np.testing.assert_allclose((gp - gm) / (2 * eps), H[:, i], atol=1e-6)
    # without the PathReferenceVelocity module the reference refuses (contouring.py:61-62)
    mm = P.ModuleManager(); mm.add_module(L.ContouringModule(st))
    with pytest.raises(IOError, match="no PathReferenceVelocity module"):
        emit.generate(mm, P.UnicycleContouringModel(), st, "bad", method="jets")


def test_plugin_base_classes_like_the_reference_tests():
    """solver_generator/test/test_base_classes.py:14-26 (Parameters) and :52-84 (model), on this repo's protocol classes."""
    from mpc_planner_amd.codegen import plugin as P
    params = P.Parameters()
    params.add("var"); params.add("v2"); params.add("long variable name")
    assert params.length() == 3
    params.load([3.8, 2.5, -1.0])
    assert params.get("var") == 3.8 and params.get("v2") == 2.5 and params.get("long variable name") == -1.0
    assert params.as_dict()["num parameters"] == 3

    for model in (P.UnicycleContouringModel(), P.UnicycleContouringSlackModel()):
        assert model.nx > 0 and model.nu > 0 and model.get_nvar() == model.nx + model.nu
        dx = model.continuous_model([0] * model.nx, [0, 0])
        assert len(dx) == model.nx and dx[0] == 0.0 and dx[1] == 0.0
        assert "x" in model.states and "y" in model.states
        model.load([1., 2., 3., 4., 5., 6., 7., 8.][:model.get_nvar()])
        assert model.get("x") == 3. and model.get("y") == 4. and model.get("a") == 1.
        with pytest.raises(IOError):
            model.get("xyz")
        lb, ub, x_range = model.get_bounds("x")
        assert ub > lb and x_range > 0
        assert model.get_bounds("w") == (-0.8, 0.8, 1.6)
    assert list(P.UnicycleContouringModel().get_xinit()) == [2, 3, 4, 5, 6]
    assert list(P.UnicycleContouringSlackModel().get_xinit()) == [2, 3, 4, 5, 6]       # slack excluded (solver_model.py:297-298)
    # unicycle at psi = pi/2, v = 2: moves along +y
    dx = P.UnicycleContouringModel().continuous_model([0., 0., np.pi / 2, 2.0, 0.], [0.5, -0.1])
    np.testing.assert_allclose(dx, [0.0, 2.0, -0.1, 0.5, 2.0], atol=1e-15)


def test_generated_model_map_carries_the_models_bounds(tmp_path):
    """Round-1 advisor finding: a plugin model's set_bounds(...) never reached the generated solver's model_map.yaml."""
    from mpc_planner_amd.generate_solver import _model_map
    from mpc_planner_amd.codegen import plugin
    model = plugin.UnicycleContouringModel()
    lb = list(model.lower_bound); ub = list(model.upper_bound)
    lb[0], ub[0], ub[5] = -1.2, 1.2, 2.5                           # a, v
    model.set_bounds(lb, ub)
    mm = _model_map(False, model)
    assert mm["a"] == ["u", 0, -1.2, 1.2] and mm["v"][3] == 2.5 and mm["w"] == ["u", 1, -0.8, 0.8]
    assert _model_map(True)["slack"] == ["x", 7, 0.0, 5000.0]
    # SecondOrderUnicycleModel (round 5): the map has the model's six variables and bounds, no `spline` row (the kernels' padding slot has no name)
    so = _model_map(False, plugin.SecondOrderUnicycleModel())
    assert list(so) == ["a", "w", "x", "y", "psi", "v"] and so["w"] == ["u", 1, -2.0, 2.0] and so["v"] == ["x", 5, -2.0, 3.0] and so["x"][2:] == [-200.0, 200.0]


def test_generated_goal_stack_on_second_order_unicycle_matches_reference_golden():
    """a stack on the reference's model WITHOUT a spline state (SecondOrderUnicycleModel, solver_model.py:170-191) goes through the
    generator: the emitted header is marked (tmpc_gen::MODEL = 1: the kernels' fifth state slot is inert), carries the model's bounds, and its
    stage functions equal the reference's own (tests/golden/stage_functions_goal.json) on the six real variables with nothing on the padding slot."""
    st = stacks.settings(N=20, max_obstacles=4)
    model, mm = stacks.goal_ellipsoids_second_order_unicycle(st)
    assert model.nx == 4 and model.get_nvar() == 6
    for method in ("jets", "symbolic"):
        gen = emit.generate(mm, model, st, "goal_so", method=method)
        assert gen["model"] == 1 and gen["npar"] == 37 and gen["nh"] == 4 and gen["slack"] == 0
        assert "constexpr int MODEL = 1;" in gen["header"] and "constexpr double LB[7] = {-2.0, -2.0, -200.0, -200.0," in gen["header"]
        with open(os.path.join(HERE, "golden", "stage_functions_goal.json")) as fh:
            gold = json.load(fh)
        assert dict(gen["params"]._params) == gold["cases"][0]["parameter_map"]
        hs = HostStageFunctions(gen["header"])
        for case in gold["cases"]:
            for pad in (0.0, 2.5):
                z = np.array(case["z"] + [pad])
                v, g, H = hs.cost(z, case["p"])
                np.testing.assert_allclose(v, case["cost"], rtol=1e-12)
                np.testing.assert_allclose(g[:6], case["cost_grad"], rtol=1e-10, atol=1e-11)
                np.testing.assert_allclose(H[:6, :6], case["cost_hess"], rtol=1e-9, atol=1e-10)
                assert g[6] == 0.0 and not H[6].any()
                h, D, Hr = hs.rows(z, case["p"])
                np.testing.assert_allclose(h, 1.0 - np.array(case["h"]), rtol=1e-11, atol=1e-12)          # ellipsoid rows h >= 1 as 1 - h <= 0
                np.testing.assert_allclose(-D, np.array(case["h_jac"])[:, 2:5], rtol=1e-10, atol=1e-11)
    # a model the kernels cannot integrate is still refused
    class Bicycle(type(model)):
        def __init__(self):
            super().__init__(); self.states = ["x", "y", "psi", "v", "delta", "spline"]; self.inputs = ["a", "w", "slack"]; self.nu, self.nx = 3, 6
    with pytest.raises(emit.UnsupportedStack):
        emit.generate(mm, Bicycle(), st, "bicycle", method="jets")
