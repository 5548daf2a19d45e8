#!/usr/bin/env python3
"""Golden vectors for the CURVATURE-AWARE contouring cost (BASELINE configs[2] "Jackal CA-MPC + decomp_util static constraints"),
made like make_golden.py / make_golden_slack.py: by executing the reference's own python modules under the sympy-backed casadi
stand-in of make_golden.py.

Reference code paths executed (unmodified, imported from /root/reference):
  curvature_aware_contouring.py:48-105    CurvatureAwareContouringObjective.get_value  (stage_idx = 1: the acados stage cost)
  spline.py:4-86                          Spline2D.at / deriv_normalized / deriv2 (deriv2 = the lambda blend of the segments' second derivatives)
  mpc_base.py:47-60                       weigh_variable(a, w, slack, v)
  solver_model.py:274-298                 ContouringSecondOrderUnicycleModelWithSlack
  guidance_constraints.py, ellipsoid_constraints.py, decomp_constraints.py      (the rows; identical to cfg3_rosnav_tmpc)

Configuration (route 1): the reference's own CA model class needs Forces-style discrete dynamics for the
spline state and is rejected by its acados path (solver_model.py:217-221); the CA *cost* runs under acados with the standard spline
ODE s' = v.  The module list is rosnavigation's configuration_tmpc (generate_rosnavigation_solver.py:86-108) with ContouringModule
replaced by CurvatureAwareContouringModule -- what its C++ side expects (curvature_aware_contouring.cpp:15-49 sets `contour`,
`terminal_*` and the spline rows and leaves `velocity` / `reference_velocity` to MPCBaseModule): the parameter map is the MPCC stack's.
  cfg3_ca_tmpc          slack model, MPCBase(a,w,slack,v), CurvatureAwareContouring, Guidance(Ellipsoid), Decomp   (N = 30, npar 172)
  ca_no_slack           ContouringSecondOrderUnicycleModel, MPCBase(a,w,v), CurvatureAwareContouring, Guidance(Ellipsoid)  (N = 20, npar 135)

Output (committed): tests/golden/stage_functions_ca.json        Run: python tests/golden/make_golden_ca.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # installs the casadi stand-in and the reference import paths  # noqa: E402
import make_golden_slack as mgs  # noqa: E402

import numpy as np  # noqa: E402
import sympy as sp  # noqa: E402

from util.parameters import Parameters  # noqa: E402
from control_modules import ModuleManager  # noqa: E402
from solver_definition import define_parameters, objective, constraints, constraint_number  # noqa: E402
from solver_model import ContouringSecondOrderUnicycleModelWithSlack, ContouringSecondOrderUnicycleModel  # noqa: E402
from mpc_base import MPCBaseModule  # noqa: E402
from curvature_aware_contouring import CurvatureAwareContouringModule  # noqa: E402
from ellipsoid_constraints import EllipsoidConstraintModule  # noqa: E402
from guidance_constraints import GuidanceConstraintModule  # noqa: E402
from decomp_constraints import DecompConstraintModule  # noqa: E402

PREC = mg.PREC


def configuration_ca_tmpc(settings):
    modules = ModuleManager()
    model = ContouringSecondOrderUnicycleModelWithSlack()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration")
    base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="slack", weight_names="slack")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(CurvatureAwareContouringModule(settings))
    modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=EllipsoidConstraintModule))
    modules.add_module(DecompConstraintModule(settings))
    return model, modules


def configuration_ca_no_slack(settings):
    modules = ModuleManager()
    model = ContouringSecondOrderUnicycleModel()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration")
    base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(CurvatureAwareContouringModule(settings))
    modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=EllipsoidConstraintModule))
    return model, modules


def fill_params_no_slack(params, M, S, rng, variant):
    """make_golden_slack.fill_params without the slack-model entries."""
    class _P:                                   # fill_params writes p[names[name]]: give it a map that swallows the names this stack lacks
        pass
    names = dict(params._params)
    sink = params.length()
    for k in ("slack",):
        names.setdefault(k, sink)
    shadow = type("S", (), {"_params": names, "length": lambda self: sink + 1})()
    p = mgs.fill_params(shadow, M, S, rng, variant, 0, 0, (0.0, 0.0))
    return p[:sink]


def main():
    out = {"_doc": "curvature-aware contouring golden vectors (tests/golden/make_golden_ca.py, reference python under a sympy casadi stand-in); "
                   "z=[a,w,x,y,psi,v,spline(,slack)]; cost = objective(..., stage_idx=1), NOT scaled by dt", "cases": []}
    cfgs = [("cfg3_ca_tmpc", configuration_ca_tmpc, 30, 8, 12, 8), ("ca_no_slack", configuration_ca_no_slack, 20, 8, 0, 7)]
    for name, conf, N, M, n_dec, nz in cfgs:
        zs = [sp.Symbol(f"z{i}", real=True) for i in range(nz)]
        settings = mg.base_settings(N, M)
        settings["decomp"] = {"range": 2.0, "max_constraints": 12}
        model, modules = conf(settings)
        params = Parameters()
        define_parameters(modules, params, settings)
        settings["params"] = params
        npar = params.length()
        nh = constraint_number(modules)
        for variant in range(3):
            rng = np.random.default_rng(9500 + 100 * nh + variant)
            s_val = [2.6, 11.97, 17.3][variant]          # (11.97: 3 cm before a knot -> the sigmoid glue and its derivatives are live)
            zval = [rng.uniform(-1.5, 1.5), rng.uniform(-0.7, 0.7), s_val + rng.uniform(-0.5, 0.5), rng.uniform(-1.0, 1.0),
                    rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5), s_val] + ([0.0 if variant == 0 else 0.37] if nz == 8 else [])
            p = mgs.fill_params(params, M, 5, rng, variant, n_dec, 0, zval[2:4]) if nz == 8 else fill_params_no_slack(params, M, 5, rng, variant)
            if variant == 2:                              # a curvy path: larger second derivatives, projection ratio well away from 1
                pm = params._params
                for i in range(5):
                    p[pm[f"spline_y{i}_b"]] = 0.12 * (-1) ** i; p[pm[f"spline_y{i}_a"]] = -0.015 * (-1) ** i
                    p[pm[f"spline_x{i}_b"]] = -0.03
            subs = {zs[i]: sp.Float(repr(zval[i]), PREC) for i in range(nz)}
            pl = [sp.Float(repr(float(v)), PREC) for v in p]
            cost = mg.scalarize(objective(modules, zs, pl, model, settings, 1))
            cg, cH = mg.grad_hess(cost, zs, subs)
            hs = [mg.scalarize(c) for c in constraints(modules, zs, pl, model, settings, 1)]
            out["cases"].append({"name": f"{name}_v{variant}", "config": name, "N": N, "M": M, "n_lin": M, "n_dec": n_dec, "S": 5, "slack": int(nz == 8),
                                 "npar": npar, "nh": nh, "parameter_map": dict(params._params), "z": zval, "p": [float(v) for v in p],
                                 "cost": mg.num(cost, subs), "cost_grad": cg, "cost_hess": cH, "h": [mg.num(h, subs) for h in hs]})
            print(name, variant, "cost", out["cases"][-1]["cost"], "npar", npar, "nh", nh, flush=True)
    with open(os.path.join(HERE, "stage_functions_ca.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote stage_functions_ca.json")


if __name__ == "__main__":
    main()
