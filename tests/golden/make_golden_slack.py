#!/usr/bin/env python3
"""Golden vectors for the SLACK-model stage functions (BASELINE configs 3 and 5), made like make_golden.py: by
executing the reference's own python modules under the sympy-backed casadi stand-in of make_golden.py.

Reference code paths executed (unmodified, imported from /root/reference):
  solver_model.py:274-298                 ContouringSecondOrderUnicycleModelWithSlack (nx = 6: ..., spline, slack)
  mpc_base.py:47-60                       weigh_variable(a, w, slack, v)
  contouring.py:48-98, spline.py          contouring cost
  guidance_constraints.py:95-110, ellipsoid_constraints.py:65-119
  decomp_constraints.py:68-98             a1 x + a2 y - (b + slack) <= 0   (12 rows)
  scenario_constraints.py:64-94           a1 x + a2 y - (b + slack) <= 0   (24 rows)
Configurations (module lists are the reference's own):
  cfg3_rosnav_tmpc      generate_rosnavigation_solver.py:86-108  configuration_tmpc: slack model, MPCBase(a,w,slack,v),
                        Contouring, Guidance(Ellipsoid), Decomp                      (N = 30, 8 obstacles, npar 172)
  cfg5_safe_horizon     generate_jackalsimulator_solver.py:67-90 configuration_safe_horizon: slack model,
                        MPCBase(a,w,slack,v), Contouring, Scenario                   (N = 20, npar 127)
  rosnav_safe_horizon   generate_rosnavigation_solver.py:62-84: ... Scenario, Decomp (npar 163)
BASELINE config 3 names "CA-MPC": the curvature-aware model is rejected by the reference's acados path
(solver_model.py:221), so the acados-runnable rosnavigation T-MPC configuration above stands in for it.

Output (committed): tests/golden/stage_functions_slack.json        Run: python tests/golden/make_golden_slack.py
"""
import json
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # installs the casadi stand-in and the reference import paths  # noqa: E402

import numpy as np  # noqa: E402
import sympy as sp  # noqa: E402

from util.parameters import Parameters  # noqa: E402
from control_modules import ModuleManager  # noqa: E402
from solver_definition import (  # noqa: E402
    define_parameters, objective, constraints, constraint_lower_bounds, constraint_upper_bounds, constraint_number)
from solver_model import ContouringSecondOrderUnicycleModelWithSlack  # noqa: E402
from mpc_base import MPCBaseModule  # noqa: E402
from contouring import ContouringModule  # noqa: E402
from ellipsoid_constraints import EllipsoidConstraintModule  # noqa: E402
from guidance_constraints import GuidanceConstraintModule  # noqa: E402
from decomp_constraints import DecompConstraintModule  # noqa: E402
from scenario_constraints import ScenarioConstraintModule  # noqa: E402

PREC = mg.PREC


def _base(settings):
    modules = ModuleManager()
    model = ContouringSecondOrderUnicycleModelWithSlack()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration")
    base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="slack", weight_names="slack")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"],
                        cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(ContouringModule(settings))
    return model, modules


def configuration_rosnav_tmpc(settings):
    model, modules = _base(settings)
    modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=EllipsoidConstraintModule))
    modules.add_module(DecompConstraintModule(settings))
    return model, modules


def configuration_safe_horizon(settings):
    model, modules = _base(settings)
    modules.add_module(ScenarioConstraintModule(settings))
    return model, modules


def configuration_rosnav_safe_horizon(settings):
    model, modules = _base(settings)
    modules.add_module(ScenarioConstraintModule(settings))
    modules.add_module(DecompConstraintModule(settings))
    return model, modules


def fill_params(params, M, S, rng, variant, n_dec, n_scen, z_xy):
    p = np.zeros(params.length())
    names = params._params
    setp = lambda name, v: p.__setitem__(names[name], v)
    for k, v in dict(acceleration=0.34, angular_velocity=0.85, slack=10000.0, velocity=0.55, reference_velocity=2.0,
                     contour=0.05, lag=0.75, terminal_angle=100.0, terminal_contouring=10.0).items():
        setp(k, v)
    amp = rng.uniform(0.2, 1.0)
    for i in range(S):
        setp(f"spline_x{i}_a", rng.uniform(-2e-3, 2e-3)); setp(f"spline_x{i}_b", rng.uniform(-1e-2, 1e-2))
        setp(f"spline_x{i}_c", 1.0 + rng.uniform(-0.05, 0.05)); setp(f"spline_x{i}_d", 6.0 * i)
        setp(f"spline_y{i}_a", amp * rng.uniform(-5e-3, 5e-3)); setp(f"spline_y{i}_b", amp * rng.uniform(-3e-2, 3e-2))
        setp(f"spline_y{i}_c", amp * rng.uniform(-0.2, 0.2)); setp(f"spline_y{i}_d", amp * rng.uniform(-0.5, 0.5))
        setp(f"spline{i}_start", 6.0 * i)
    if "ego_disc_radius" in names:
        setp("ego_disc_radius", 0.325)
    setp("ego_disc_0_offset", 0.0 if variant % 2 == 0 else 0.12)
    obs = []
    for j in range(M):
        ox, oy = rng.uniform(1.0, 16.0), rng.uniform(-4.0, 4.0)
        obs.append((ox, oy))
        setp(f"ellipsoid_obst_{j}_x", ox); setp(f"ellipsoid_obst_{j}_y", oy)
        setp(f"ellipsoid_obst_{j}_psi", rng.uniform(-math.pi, math.pi))
        gaussian = (j % 2 == 1)
        setp(f"ellipsoid_obst_{j}_major", rng.uniform(0.1, 0.8) if gaussian else 0.0)
        setp(f"ellipsoid_obst_{j}_minor", rng.uniform(0.05, 0.4) if gaussian else 0.0)
        setp(f"ellipsoid_obst_{j}_chi", 5.991464547107979 if gaussian else 1.0)
        setp(f"ellipsoid_obst_{j}_r", 0.4)
    if M:
        g = np.array([rng.uniform(0.0, 6.0), rng.uniform(-1.0, 1.0)])
        for j in range(M):
            o = np.array(obs[j]); a = (o - g) / np.linalg.norm(o - g)
            setp(f"lin_constraint_{j}_a1", a[0]); setp(f"lin_constraint_{j}_a2", a[1])
            setp(f"lin_constraint_{j}_b", a @ o - (1e-3 + 0.325))
    c = np.array(z_xy)
    for j in range(n_dec):          # corridor polytope around the point (decomp_constraints.cpp:90-113 semantics)
        th = 2.0 * math.pi * (j + rng.uniform(-0.3, 0.3)) / n_dec
        a = np.array([math.cos(th), math.sin(th)])
        setp(f"disc_0_decomp_{j}_a1", a[0]); setp(f"disc_0_decomp_{j}_a2", a[1])
        setp(f"disc_0_decomp_{j}_b", a @ c + rng.uniform(0.2, 2.5))
    for j in range(n_scen):
        th = 2.0 * math.pi * rng.uniform(0.0, 1.0)
        a = np.array([math.cos(th), math.sin(th)])
        setp(f"disc_0_scenario_constraint_{j}_a1", a[0]); setp(f"disc_0_scenario_constraint_{j}_a2", a[1])
        setp(f"disc_0_scenario_constraint_{j}_b", a @ c + rng.uniform(-0.1, 3.0))
    return p


def erk4(model, x, u, dt, num_steps):
    h = sp.Rational(1, num_steps) * dt
    f = lambda xx: [sp.sympify(e) for e in model.continuous_model(xx, u)]
    for _ in range(num_steps):
        k1 = f(x)
        k2 = f([xi + h / 2 * ki for xi, ki in zip(x, k1)])
        k3 = f([xi + h / 2 * ki for xi, ki in zip(x, k2)])
        k4 = f([xi + h * ki for xi, ki in zip(x, k3)])
        x = [xi + h / 6 * (a + 2 * b + 2 * c + d) for xi, a, b, c, d in zip(x, k1, k2, k3, k4)]
    return x


def main():
    out = {"_doc": "slack-model golden vectors (tests/golden/make_golden_slack.py, reference python under a sympy casadi "
                   "stand-in); z=[a,w,x,y,psi,v,spline,slack]", "cases": []}
    zs = [sp.Symbol(f"z{i}", real=True) for i in range(8)]
    cfgs = [("cfg3_rosnav_tmpc", configuration_rosnav_tmpc, 30, 8, 12, 0),
            ("cfg5_safe_horizon", configuration_safe_horizon, 20, 0, 0, 24),
            ("rosnav_safe_horizon", configuration_rosnav_safe_horizon, 20, 0, 12, 24)]
    for name, conf, N, M, n_dec, n_scen in cfgs:
        settings = mg.base_settings(N, max(M, 1))
        settings["decomp"] = {"range": 2.0, "max_constraints": 12}
        model, modules = conf(settings)
        params = Parameters()
        define_parameters(modules, params, settings)
        settings["params"] = params
        npar = params.length()
        nh = constraint_number(modules)
        lb = [(-1e15 if v == -np.inf else v) for v in constraint_lower_bounds(modules)]
        ub = [(1e15 if v == np.inf else v) for v in constraint_upper_bounds(modules)]
        for variant in range(2):
            rng = np.random.default_rng(9000 + 100 * nh + variant)
            s_val = [2.6, 11.97][variant]
            zval = [rng.uniform(-1.5, 1.5), rng.uniform(-0.7, 0.7), s_val + rng.uniform(-0.5, 0.5),
                    rng.uniform(-1.0, 1.0), rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5), s_val,
                    0.0 if variant == 0 else 0.37]      # slack = 0 is what the acados path always sees (DESIGN U9)
            p = fill_params(params, M, 5, rng, variant, n_dec, n_scen, zval[2:4])
            subs = {zs[i]: sp.Float(repr(zval[i]), PREC) for i in range(8)}
            pl = [sp.Float(repr(float(v)), PREC) for v in p]
            # the model's own variable lookup (model.get) indexes z by name: load z the way solver_definition does
            cost = mg.scalarize(objective(modules, zs, pl, model, settings, 1))
            cg, cH = mg.grad_hess(cost, zs, subs)
            hs = [mg.scalarize(c) for c in constraints(modules, zs, pl, model, settings, 1)]
            assert len(hs) == nh
            hval, hjac, hhess = [], [], []
            for hexpr in hs:
                hval.append(mg.num(hexpr, subs))
                g, H = mg.grad_hess(hexpr, zs, subs)
                hjac.append(g); hhess.append(H)
            f = [sp.sympify(e) for e in model.continuous_model(zs[2:], zs[:2])]
            xn = erk4(model, zs[2:], zs[:2], sp.Float("0.2", PREC), 3)
            dval, djac, dhess = [], [], []
            for e in xn:
                dval.append(mg.num(e, subs))
                g, H = mg.grad_hess(e, zs, subs)
                djac.append(g); dhess.append(H)
            out["cases"].append({
                "name": f"{name}_v{variant}", "config": name, "N": N, "M": M, "n_lin": M, "n_dec": n_dec, "n_scen": n_scen,
                "S": 5, "npar": npar, "nh": nh, "parameter_map": dict(params._params), "lh": lb, "uh": ub,
                "lower_bound": list(model.lower_bound), "upper_bound": list(model.upper_bound),
                "z": zval, "p": [float(v) for v in p],
                "cost": mg.num(cost, subs), "cost_grad": cg, "cost_hess": cH,
                "h": hval, "h_jac": hjac, "h_hess": hhess,
                "f_cont": [mg.num(e, subs) for e in f],
                "x_next": dval, "x_next_jac": djac, "x_next_hess": dhess})
            print(name, variant, "cost", out["cases"][-1]["cost"], "npar", npar, "nh", nh, flush=True)
    with open(os.path.join(HERE, "stage_functions_slack.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote stage_functions_slack.json")


if __name__ == "__main__":
    main()
