"""Drop-in check of the generator's plugin surface: the reference's OWN, unmodified module scripts
(/root/reference/mpc_planner_modules/scripts, solver_generator) are imported with mpc_planner_amd.codegen.symbolic
registered as `casadi`, assembled exactly like generate_jackalsimulator_solver.py:92-101 does, and handed to emit.generate.
Only runs where the reference checkout exists (this container); a subprocess with bytecode writing disabled keeps the
checkout untouched."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
from mpc_planner_amd.codegen import symbolic, emit, cpp_glue
symbolic.install_as_casadi()
sys.path.insert(0, %(ref)r + "/solver_generator"); sys.path.insert(0, %(ref)r + "/mpc_planner_modules/scripts")
from control_modules import ModuleManager
from solver_model import ContouringSecondOrderUnicycleModel
from mpc_base import MPCBaseModule
from contouring import ContouringModule
from ellipsoid_constraints import EllipsoidConstraintModule
from guidance_constraints import GuidanceConstraintModule
settings = {"N": 20, "integrator_step": 0.2, "n_discs": 1, "max_obstacles": 8, "linearized_constraints": {"add_halfspaces": 0},
            "contouring": {"num_segments": 5, "dynamic_velocity_reference": False}}
modules = ModuleManager(); model = ContouringSecondOrderUnicycleModel()
base = modules.add_module(MPCBaseModule(settings))
base.weigh_variable(var_name="a", weight_names="acceleration"); base.weigh_variable(var_name="w", weight_names="angular_velocity")
base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
modules.add_module(ContouringModule(settings))
modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=EllipsoidConstraintModule))
gen = emit.generate(modules, model, settings, "reference_scripts_tmpc", method="jets")
# second configuration: the reference's own generation test stack (solver_generator/test/test_acados.py:30-46)
from path_reference_velocity import PathReferenceVelocityModule
from solver_definition import define_parameters
from util.parameters import Parameters
st2 = {"N": 20, "integrator_step": 0.2, "n_discs": 1, "max_obstacles": 12, "contouring": {"num_segments": 8, "dynamic_velocity_reference": False}}
m2 = ModuleManager()
b2 = m2.add_module(MPCBaseModule(st2))
b2.weigh_variable(var_name="a", weight_names="acceleration"); b2.weigh_variable(var_name="w", weight_names="angular_velocity")
m2.add_module(ContouringModule(st2)); m2.add_module(PathReferenceVelocityModule(st2)); m2.add_module(EllipsoidConstraintModule(st2))
p2 = define_parameters(m2, Parameters(), st2)
json.dump(dict(header=gen["header"], pmap=dict(gen["params"]._params), nh=gen["nh"], modules_h=cpp_glue.modules_header(modules),
               pmap2=dict(p2._params), modules_h2=cpp_glue.modules_header(m2), definitions_h2=cpp_glue.definitions_header(m2),
               definitions_h=cpp_glue.definitions_header(modules), modules_cmake=cpp_glue.modules_cmake(modules)), open(sys.argv[1], "w"))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "solver_generator")), reason="reference checkout not present")
def test_unmodified_reference_module_scripts_generate_the_same_stage_functions(tmp_path):
    out = tmp_path / "gen.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, ref=REF), str(out)], check=True, env=env, timeout=600,
                   cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    gen = json.load(open(out))
    from mpc_planner_amd.codegen.hostlib import HostStageFunctions
    with open(os.path.join(HERE, "golden", "stage_functions.json")) as fh:
        cases = [c for c in json.load(fh)["cases"] if c["config"] == "cfg2_tmpc_M8"]
    assert gen["pmap"] == cases[0]["parameter_map"] and gen["nh"] == 16
    hs = HostStageFunctions(gen["header"])
    for c in cases:
        v, g, H = hs.cost(c["z"], c["p"])
        np.testing.assert_allclose(v, c["cost"], rtol=1e-12)
        np.testing.assert_allclose(g, c["cost_grad"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(H, c["cost_hess"], rtol=1e-9, atol=1e-10)
        h, D, Hr = hs.rows(c["z"], c["p"])
        hh, J = np.array(c["h"]), np.array(c["h_jac"])
        for k in range(hs.nh):
            r, s = hs.row_src[k], hs.row_sign[k]
            np.testing.assert_allclose(h[k], s * (hh[r] - hs.row_bound[k]), rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(D[k], s * J[r][[2, 3, 4]], rtol=1e-10, atol=1e-11)
    # the C++ wiring files (modules.h / definitions.h / modules.cmake) written for the reference's own module objects are
    # identical to those written for this repo's library stack of the same configuration
    from mpc_planner_amd.codegen import cpp_glue, stacks
    _, mm = stacks.tmpc(stacks.settings())
    assert gen["modules_h"] == cpp_glue.modules_header(mm)
    assert gen["definitions_h"] == cpp_glue.definitions_header(mm)
    assert gen["modules_cmake"] == cpp_glue.modules_cmake(mm)
    # the stack of the reference's generation test (test_acados.py:30-46): same parameter map and wiring files from the
    # reference's own Parameters / module objects and from this repo's library
    from mpc_planner_amd.codegen import plugin as P
    st2 = stacks.settings(N=20, max_obstacles=12, num_segments=8)
    _, mm2 = stacks.contouring_path_velocity_ellipsoids(st2)
    assert gen["pmap2"] == dict(P.define_parameters(mm2, P.Parameters(), st2)._params)
    assert gen["modules_h2"] == cpp_glue.modules_header(mm2) and gen["definitions_h2"] == cpp_glue.definitions_header(mm2)


ALL_MODULES_SCRIPT = r'''
import sys, json, inspect
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
from mpc_planner_amd.codegen import symbolic
symbolic.install_as_casadi()
sys.path.insert(0, %(ref)r + "/solver_generator"); sys.path.insert(0, %(ref)r + "/mpc_planner_modules/scripts")
import numpy as np, sympy as sp
import solver_model
from control_modules import ModuleManager
from solver_definition import define_parameters, objective, constraints, constraint_lower_bounds, constraint_upper_bounds, constraint_number
from util.parameters import Parameters
from mpc_base import MPCBaseModule
from contouring import ContouringModule
from curvature_aware_contouring import CurvatureAwareContouringModule
from goal_module import GoalModule
from path_reference_velocity import PathReferenceVelocityModule
from ellipsoid_constraints import EllipsoidConstraintModule
from gaussian_constraints import GaussianConstraintModule
from guidance_constraints import GuidanceConstraintModule
from linearized_constraints import LinearizedConstraintModule
from scenario_constraints import ScenarioConstraintModule
from decomp_constraints import DecompConstraintModule
settings = dict(n_discs=1, max_obstacles=1, N=20, contouring=dict(num_segments=10, dynamic_velocity_reference=False),
                linearized_constraints=dict(add_halfspaces=2), decomp=dict(range=2.0, max_constraints=4))
modules = ModuleManager()                                   # test_control_modules.py:106-136 (+ the decomp module)
b = modules.add_module(MPCBaseModule(settings))
b.weigh_variable(var_name="a", weight_names="acceleration"); b.weigh_variable(var_name="w", weight_names="angular_velocity")
b.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
for M in (GoalModule, ContouringModule, CurvatureAwareContouringModule, PathReferenceVelocityModule, GaussianConstraintModule,
          EllipsoidConstraintModule, GuidanceConstraintModule, LinearizedConstraintModule, ScenarioConstraintModule, DecompConstraintModule):
    modules.add_module(M(settings))
params = define_parameters(modules, Parameters(), settings); settings["params"] = params
model = solver_model.ContouringSecondOrderUnicycleModelWithSlack()
z = np.array([sp.Symbol(f"z{i}", real=True) for i in range(model.get_nvar())], dtype=object)
p = [sp.Symbol(f"p{i}", real=True) for i in range(params.length())]
cost = objective(modules, z, p, model, settings, 1)
rows = constraints(modules, z, p, model, settings, 1)
lb, ub = constraint_lower_bounds(modules), constraint_upper_bounds(modules)
out = dict(names=[m.module_name for m in modules.modules], npar=params.length(), n_rows=len(rows), n_lb=len(lb), n_ub=len(ub),
           num=constraint_number(modules), cost_symbols=len(sp.sympify(cost).free_symbols))
# every shipped model traces through the facade (solver_model.py:170-437)
models = {}
for name, cls in inspect.getmembers(solver_model, inspect.isclass):
    if cls.__module__ != "solver_model" or not hasattr(cls, "continuous_model") or name == "DynamicsModel":
        continue
    m = cls()
    zz = np.array([sp.Symbol(f"z{i}", real=True) for i in range(m.get_nvar())], dtype=object)
    m.load(zz)
    dx = m.continuous_model(zz[m.nu:], zz[:m.nu])
    models[name] = [m.nx, m.nu, int(np.asarray(dx, dtype=object).size)]
out["models"] = models
json.dump(out, open(sys.argv[1], "w"))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "solver_generator")), reason="reference checkout not present")
def test_every_shipped_module_script_and_model_traces_through_the_facade(tmp_path):
    """SURVEY 8 f-4 asks for a CasADi-compatible tracer for all shipped module scripts and models: the reference's own
    `test_all_modules` stack (test_control_modules.py:106-136, plus the decomp module) is constructed from the unmodified
    scripts, its objective and constraints are evaluated symbolically, and every model class of solver_model.py evaluates
    its continuous dynamics -- all with mpc_planner_amd.codegen.symbolic standing in for `casadi`."""
    out = tmp_path / "all.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", ALL_MODULES_SCRIPT % dict(root=ROOT, ref=REF), str(out)], check=True, env=env, timeout=600,
                   cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    r = json.load(open(out))
    assert r["names"] == ["MPCBaseModule", "GoalModule", "Contouring", "CurvatureAwareContouring", "PathReferenceVelocity",
                          "GaussianConstraints", "EllipsoidConstraints", "GuidanceConstraints", "LinearizedConstraints",
                          "ScenarioConstraints", "DecompConstraints"]
    assert r["n_rows"] == r["n_lb"] == r["n_ub"] == r["num"] == 1 + 1 + (3 + 1) + 1 + 24 + 4
    assert r["npar"] == 252 and r["cost_symbols"] > 20
    assert len(r["models"]) >= 5
    for name, (nx, nu, n_dx) in r["models"].items():
        assert n_dx in (nx, nx - 1) and nu >= 2, name       # curvature-aware models integrate all but the last state
    assert r["models"]["ContouringSecondOrderUnicycleModel"] == [5, 2, 5]
    assert r["models"]["ContouringSecondOrderUnicycleModelWithSlack"] == [6, 2, 6]


CAC_SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
from mpc_planner_amd.codegen import symbolic, emit
symbolic.install_as_casadi()
sys.path.insert(0, %(ref)r + "/solver_generator"); sys.path.insert(0, %(ref)r + "/mpc_planner_modules/scripts")
from control_modules import ModuleManager
from solver_model import ContouringSecondOrderUnicycleModel
from mpc_base import MPCBaseModule
from curvature_aware_contouring import CurvatureAwareContouringModule
from linearized_constraints import LinearizedConstraintModule
settings = dict(n_discs=1, max_obstacles=2, N=20, integrator_step=0.2, contouring=dict(num_segments=3, dynamic_velocity_reference=False),
                linearized_constraints=dict(add_halfspaces=1))
modules = ModuleManager()
b = modules.add_module(MPCBaseModule(settings))
b.weigh_variable(var_name="a", weight_names="acceleration"); b.weigh_variable(var_name="w", weight_names="angular_velocity")
modules.add_module(CurvatureAwareContouringModule(settings)); modules.add_module(LinearizedConstraintModule(settings))
gen = emit.generate(modules, ContouringSecondOrderUnicycleModel(), settings, "cac", method="jets")
json.dump(dict(header=gen["header"], pmap=dict(gen["params"]._params), nh=gen["nh"]), open(sys.argv[1], "w"))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "solver_generator")), reason="reference checkout not present")
def test_modules_without_a_library_counterpart_are_emitted_from_the_reference_scripts(tmp_path):
    """CurvatureAwareContouringModule and LinearizedConstraintModule exist only as reference scripts (no class in
    codegen/library.py, no hand-written kernel): the generator differentiates and emits them as they are; emitted
    gradients / Hessians against central differences of the emitted values."""
    out = tmp_path / "cac.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", CAC_SCRIPT % dict(root=ROOT, ref=REF), str(out)], check=True, env=env, timeout=600,
                   cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    gen = json.load(open(out))
    from mpc_planner_amd.codegen.hostlib import HostStageFunctions
    hs = HostStageFunctions(gen["header"])
    pm = gen["pmap"]
    assert hs.npar == len(pm) and hs.nh == gen["nh"] == 2 and "disc_0_lin_constraint_0_a1" in pm and "spline_x0_a" in pm
    rng = np.random.default_rng(0)
    p = rng.uniform(0.2, 1.0, hs.npar)
    for i, start in enumerate((0.0, 4.0, 8.0)):
        p[pm[f"spline{i}_start"]] = start
    z = np.array([0.3, -0.1, 1.0, 0.4, 0.2, 1.1, 2.5])
    v, g, H = hs.cost(z, p)
    eps = 1e-6
    for i in range(7):
        e = np.zeros(7); e[i] = eps
        vp, gp, _ = hs.cost(z + e, p); vm, gm, _ = hs.cost(z - e, p)
        assert abs((vp - vm) / (2 * eps) - g[i]) < 1e-6
        np.testing.assert_allclose((gp - gm) / (2 * eps), H[:, i], atol=1e-5)
    h, D, Hr = hs.rows(z, p)
    off = p[pm["ego_disc_0_offset"]]
    dx, dy = z[2] + off * np.cos(z[4]), z[3] + off * np.sin(z[4])          # disc position (linearized_constraints.py:83-86)
    for k in range(hs.nh):                                   # halfspaces a1 X_d + a2 Y_d - b <= 0
        j = hs.row_src[k]
        a1, a2, b = (p[pm[f"disc_0_lin_constraint_{j}_{f}"]] for f in ("a1", "a2", "b"))
        assert abs(h[k] - (a1 * dx + a2 * dy - b)) < 1e-12
        np.testing.assert_allclose(D[k], [a1, a2, off * (-a1 * np.sin(z[4]) + a2 * np.cos(z[4]))], atol=1e-14)
        np.testing.assert_allclose(Hr[k][2, 2], off * (-a1 * np.cos(z[4]) - a2 * np.sin(z[4])), atol=1e-14)
