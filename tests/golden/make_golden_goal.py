#!/usr/bin/env python3
"""Golden vectors for a stack on the reference's model WITHOUT a spline state: SecondOrderUnicycleModel
(solver_model.py:170-191) + MPC base (a, w, v) + GoalModule (goal_module.py:22-36) + EllipsoidConstraintModule
(ellipsoid_constraints.py:66-110).  Made like make_golden.py: by executing the reference's own python modules -- model, goal objective,
ellipsoid rows, the acados ERK4 x 3 discretisation of the model's continuous_model -- under the sympy-backed casadi stand-in.
z = [a, w, x, y, psi, v] (6 variables: the model has no fifth state).   Output (committed): tests/golden/stage_functions_goal.json"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

import numpy as np  # noqa: E402
import sympy as sp  # noqa: E402

from util.parameters import Parameters  # noqa: E402
from control_modules import ModuleManager  # noqa: E402
from solver_definition import define_parameters, objective, constraints, constraint_lower_bounds, constraint_upper_bounds  # noqa: E402
from solver_model import SecondOrderUnicycleModel  # noqa: E402
from mpc_base import MPCBaseModule  # noqa: E402
from goal_module import GoalModule  # noqa: E402
from ellipsoid_constraints import EllipsoidConstraintModule  # noqa: E402

PREC = mg.PREC
M = 4


def main():
    settings = mg.base_settings(20, M)
    modules = ModuleManager(); model = SecondOrderUnicycleModel()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration"); base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(GoalModule(settings)); modules.add_module(EllipsoidConstraintModule(settings))
    params = Parameters(); define_parameters(modules, params, settings); settings["params"] = params
    names = params._params
    lb = [(-1e15 if v == -np.inf else v) for v in constraint_lower_bounds(modules)]
    ub = [(1e15 if v == np.inf else v) for v in constraint_upper_bounds(modules)]
    nz = model.nu + model.nx
    assert nz == 6 and model.states == ["x", "y", "psi", "v"]
    zs = [sp.Symbol(f"z{i}", real=True) for i in range(nz)]
    out = {"_doc": "goal tracking + ellipsoids on SecondOrderUnicycleModel, tests/golden/make_golden_goal.py (the reference's own python); z=[a,w,x,y,psi,v]",
           "model": {"states": model.states, "inputs": model.inputs, "lower_bound": [float(v) for v in model.lower_bound], "upper_bound": [float(v) for v in model.upper_bound]},
           "cases": []}
    for variant in range(3):
        rng = np.random.default_rng(9900 + variant)
        p = np.zeros(params.length()); setp = lambda n, v: p.__setitem__(names[n], v)
        for k, v in dict(acceleration=0.34, angular_velocity=0.85, velocity=0.55, reference_velocity=2.0, goal_weight=[4.0, 1.5, 0.7][variant],
                         goal_x=[9.0, -3.0, 0.05][variant], goal_y=[0.5, 6.0, -0.02][variant], ego_disc_radius=0.325,
                         ego_disc_0_offset=[0.0, 0.12, 0.0][variant]).items():
            setp(k, v)
        for j in range(M):
            for f, v in dict(x=rng.uniform(1.0, 10.0), y=rng.uniform(-3.0, 3.0), psi=rng.uniform(-1.0, 1.0), major=rng.uniform(0.0, 0.5),
                             minor=rng.uniform(0.0, 0.3), chi=[1.0, 5.991464547107979][j % 2], r=0.4).items():
                setp(f"ellipsoid_obst_{j}_{f}", v)
        zval = [rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(0.0, 6.0), rng.uniform(-1.0, 1.0), rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5)]
        subs = {zs[i]: sp.Float(repr(zval[i]), PREC) for i in range(nz)}
        pl = [sp.Float(repr(float(v)), PREC) for v in p]
        cost = mg.scalarize(objective(modules, zs, pl, model, settings, 1))
        cg, cH = mg.grad_hess(cost, zs, subs)
        hs = [mg.scalarize(c) for c in constraints(modules, zs, pl, model, settings, 1)]
        hval, hjac, hhess = [], [], []
        for hexpr in hs:
            hval.append(mg.num(hexpr, subs)); gg, HH = mg.grad_hess(hexpr, zs, subs); hjac.append(gg); hhess.append(HH)
        f = list(model.continuous_model(zs[2:], zs[:2]))
        xn = mg.erk4(model, zs[2:], zs[:2], sp.Float("0.2", PREC), 3)
        dval, djac, dhess = [], [], []
        for e in xn:
            dval.append(mg.num(e, subs)); gg, HH = mg.grad_hess(e, zs, subs); djac.append(gg); dhess.append(HH)
        out["cases"].append({"name": f"goal_so_unicycle_v{variant}", "N": 20, "M": M, "npar": params.length(), "nh": len(hs), "parameter_map": dict(names),
                             "lh": lb, "uh": ub, "z": zval, "p": [float(v) for v in p], "cost": mg.num(cost, subs), "cost_grad": cg, "cost_hess": cH,
                             "h": hval, "h_jac": hjac, "h_hess": hhess, "f_cont": [mg.num(e, subs) for e in f],
                             "x_next": dval, "x_next_jac": djac, "x_next_hess": dhess})
        print(variant, "cost", out["cases"][-1]["cost"], "npar", params.length(), "nh", len(hs), flush=True)
    with open(os.path.join(HERE, "stage_functions_goal.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote stage_functions_goal.json")


if __name__ == "__main__":
    main()
