#!/usr/bin/env python3
"""Golden vectors for mpc_planner_jackal's default configuration (generate_jackal_solver.py:53-73 configuration_tmpc):
MPC base (a, w, v) + contouring + GuidanceConstraintModule with the **GaussianConstraintModule** as collision-avoidance
submodule (CC-MPC rows, gaussian_constraints.py:33-113).  Made like make_golden.py: by executing the reference's own
python modules under the sympy-backed casadi stand-in.  settings: mpc_planner_jackal/config/settings.yaml (5 obstacles,
3 spline segments).   Output (committed): tests/golden/stage_functions_gaussian.json"""
import json
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

import numpy as np  # noqa: E402
import sympy as sp  # noqa: E402

np.Inf = np.inf   # gaussian_constraints.py:65 uses the NumPy-1 spelling, removed in NumPy 2


# gaussian_constraints.py:116 does `a_ij.T @ cd.SX(diff_pos)` with numpy vectors: SX(1-D ndarray) must stay an ndarray
_orig_call = type(mg.SX).__call__
type(mg.SX).__call__ = lambda cls, *a: (np.array(a[0], dtype=object) if len(a) == 1 and isinstance(a[0], np.ndarray) and a[0].ndim == 1
                                        else _orig_call(cls, *a))

from util.parameters import Parameters  # noqa: E402
from control_modules import ModuleManager  # noqa: E402
from solver_definition import define_parameters, objective, constraints, constraint_lower_bounds, constraint_upper_bounds  # noqa: E402
from solver_model import ContouringSecondOrderUnicycleModel  # noqa: E402
from mpc_base import MPCBaseModule  # noqa: E402
from contouring import ContouringModule  # noqa: E402
from gaussian_constraints import GaussianConstraintModule  # noqa: E402
from guidance_constraints import GuidanceConstraintModule  # noqa: E402

PREC = mg.PREC
M, S = 5, 3


def main():
    settings = mg.base_settings(30, M, num_segments=S)
    modules = ModuleManager(); model = ContouringSecondOrderUnicycleModel()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration"); base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(ContouringModule(settings))
    modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=GaussianConstraintModule))
    params = Parameters(); define_parameters(modules, params, settings); settings["params"] = params
    names = params._params
    lb = [(-1e15 if v == -np.inf else v) for v in constraint_lower_bounds(modules)]
    ub = [(1e15 if v == np.inf else v) for v in constraint_upper_bounds(modules)]
    zs = [sp.Symbol(f"z{i}", real=True) for i in range(7)]
    out = {"_doc": "mpc_planner_jackal default T-MPC (guidance + Gaussian chance constraints), tests/golden/make_golden_gaussian.py", "cases": []}
    for variant in range(2):
        rng = np.random.default_rng(8800 + variant)
        p = np.zeros(params.length()); setp = lambda n, v: p.__setitem__(names[n], v)
        for k, v in dict(acceleration=0.34, angular_velocity=0.85, velocity=0.55, reference_velocity=2.0, contour=0.05, lag=0.75,
                         terminal_angle=100.0, terminal_contouring=10.0, ego_disc_radius=0.325,
                         ego_disc_0_offset=0.0 if variant == 0 else 0.12).items():
            setp(k, v)
        for i in range(S):
            setp(f"spline_x{i}_a", rng.uniform(-2e-3, 2e-3)); setp(f"spline_x{i}_b", rng.uniform(-1e-2, 1e-2))
            setp(f"spline_x{i}_c", 1.0 + rng.uniform(-0.05, 0.05)); setp(f"spline_x{i}_d", 6.0 * i)
            setp(f"spline_y{i}_a", rng.uniform(-5e-3, 5e-3)); setp(f"spline_y{i}_b", rng.uniform(-3e-2, 3e-2))
            setp(f"spline_y{i}_c", rng.uniform(-0.2, 0.2)); setp(f"spline_y{i}_d", rng.uniform(-0.5, 0.5))
            setp(f"spline{i}_start", 6.0 * i)
        s_val = [2.6, 11.97][variant]
        zval = [rng.uniform(-1.5, 1.5), rng.uniform(-0.7, 0.7), s_val + rng.uniform(-0.5, 0.5), rng.uniform(-1.0, 1.0),
                rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5), s_val]
        g = np.array([rng.uniform(0.0, 6.0), rng.uniform(-1.0, 1.0)])
        for j in range(M):
            o = np.array([rng.uniform(1.0, 16.0), rng.uniform(-4.0, 4.0)])
            for f, v in dict(x=o[0], y=o[1], major=rng.uniform(0.1, 0.8), minor=rng.uniform(0.05, 0.4), risk=[0.05, 0.01, 0.1, 0.05, 0.2][j], r=0.4).items():
                setp(f"gaussian_obst_{j}_{f}", v)
            a = (o - g) / np.linalg.norm(o - g)
            setp(f"lin_constraint_{j}_a1", a[0]); setp(f"lin_constraint_{j}_a2", a[1]); setp(f"lin_constraint_{j}_b", a @ o - (1e-3 + 0.325))
        subs = {zs[i]: sp.Float(repr(zval[i]), PREC) for i in range(7)}
        pl = [sp.Float(repr(float(v)), PREC) for v in p]
        cost = mg.scalarize(objective(modules, zs, pl, model, settings, 1))
        cg, cH = mg.grad_hess(cost, zs, subs)
        hs = [mg.scalarize(c) for c in constraints(modules, zs, pl, model, settings, 1)]
        hval, hjac, hhess = [], [], []
        for hexpr in hs:
            hval.append(mg.num(hexpr, subs)); gg, HH = mg.grad_hess(hexpr, zs, subs); hjac.append(gg); hhess.append(HH)
        out["cases"].append({"name": f"jackal_tmpc_gaussian_v{variant}", "N": 30, "M": M, "S": S, "npar": params.length(), "nh": len(hs),
                             "parameter_map": dict(names), "lh": lb, "uh": ub, "z": zval, "p": [float(v) for v in p],
                             "cost": mg.num(cost, subs), "cost_grad": cg, "cost_hess": cH, "h": hval, "h_jac": hjac, "h_hess": hhess})
        print(variant, "cost", out["cases"][-1]["cost"], "npar", params.length(), "nh", len(hs), flush=True)
    with open(os.path.join(HERE, "stage_functions_gaussian.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote stage_functions_gaussian.json")


if __name__ == "__main__":
    main()
