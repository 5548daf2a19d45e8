#!/usr/bin/env python3
"""Generate golden vectors for the NLP stage functions by EXECUTING THE REFERENCE'S OWN PYTHON.

The reference defines its NLP symbolically (CasADi) in
  /root/reference/solver_generator/{solver_model.py, solver_definition.py, spline.py, util/math.py}
  /root/reference/mpc_planner_modules/scripts/{mpc_base.py, contouring.py, ellipsoid_constraints.py,
                                               guidance_constraints.py}
and lets CasADi differentiate it at solver-generation time (generate_acados_solver.py:27-65,190).
CasADi is not installed here, but the module code is plain Python arithmetic on whatever objects it is
handed.  This script installs a ~60-line `casadi` stand-in backed by **sympy**, imports the reference
modules unmodified from /root/reference (read-only, bytecode writing disabled), builds the very same
stage cost / constraint / dynamics expressions the acados generator would build (stage_idx = 1,
generate_acados_solver.py:41,48), differentiates them exactly with sympy and evaluates value, gradient
and Hessian with 40-digit arithmetic, rounded once to double.

Outputs (committed): tests/golden/stage_functions.json

Only this container has /root/reference; the JSON travels, this script documents how it was made.
Run:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import sympy as sp

# ----------------------------------------------------------------------------------------------
# casadi stand-in (sympy backed).  Covers exactly the ops the reference modules use.
# ----------------------------------------------------------------------------------------------
sp.Expr.exp = lambda self: sp.exp(self)  # spline.py:37 calls np.exp() on a symbol -> obj.exp()
sp.Expr.sqrt = lambda self: sp.sqrt(self)
sp.Expr.cos = lambda self: sp.cos(self)
sp.Expr.sin = lambda self: sp.sin(self)


def _flat(args):
    out = []
    for a in args:
        if isinstance(a, sp.MatrixBase):
            out.extend(list(a))
        elif isinstance(a, (list, tuple, np.ndarray)):
            out.extend(_flat(list(a)))
        else:
            out.append(a)
    return out


class _SXMeta(type):
    def __call__(cls, *args):
        if len(args) == 2 and all(isinstance(a, int) for a in args):
            return sp.zeros(args[0], args[1])
        if len(args) == 1:
            a = args[0]
            if isinstance(a, np.ndarray):
                if a.ndim == 1:
                    return sp.Matrix(len(a), 1, list(a))
                return sp.Matrix(a.tolist())
            if isinstance(a, sp.MatrixBase):
                return a
            return sp.sympify(a)
        if len(args) == 0:
            return sp.zeros(0, 1)
        raise NotImplementedError(args)


class SX(metaclass=_SXMeta):
    @staticmethod
    def sym(name, n=1):
        if n == 1:
            return sp.Symbol(name, real=True)
        return [sp.Symbol(f"{name}_{i}", real=True) for i in range(n)]


casadi = types.ModuleType("casadi")
casadi.SX = SX
casadi.cos, casadi.sin, casadi.tan = sp.cos, sp.sin, sp.tan
casadi.sqrt, casadi.exp, casadi.log = sp.sqrt, sp.exp, sp.log
casadi.atan2, casadi.atan, casadi.arctan = sp.atan2, sp.atan, sp.atan
casadi.erf = sp.erf
casadi.fabs = sp.Abs
casadi.fmax = lambda a, b: sp.Max(a, b)
casadi.fmod = lambda a, b: a - b * sp.floor(a / b)  # unused at stage_idx=1
casadi.pi = sp.pi
casadi.vertcat = lambda *a: _flat(a)
sys.modules["casadi"] = casadi

sys.path.insert(0, os.path.join(REF, "solver_generator"))
sys.path.insert(0, os.path.join(REF, "mpc_planner_modules", "scripts"))

# reference imports (unmodified reference code)
from util.parameters import Parameters  # noqa: E402
from control_modules import ModuleManager  # noqa: E402
from solver_definition import (  # noqa: E402
    define_parameters, objective, constraints, constraint_lower_bounds, constraint_upper_bounds,
    constraint_number,
)
from solver_model import ContouringSecondOrderUnicycleModel  # noqa: E402
from mpc_base import MPCBaseModule  # noqa: E402
from contouring import ContouringModule  # noqa: E402
from ellipsoid_constraints import EllipsoidConstraintModule  # noqa: E402
from guidance_constraints import GuidanceConstraintModule  # noqa: E402

PREC = 40


def base_settings(N, max_obstacles, num_segments=5):
    # mirrors mpc_planner_jackalsimulator/config/settings.yaml (only keys the modules read)
    return {
        "N": N,
        "integrator_step": 0.2,
        "n_discs": 1,
        "max_obstacles": max_obstacles,
        "linearized_constraints": {"add_halfspaces": 0},
        "contouring": {"num_segments": num_segments, "dynamic_velocity_reference": False},
    }


def configuration_no_obstacles(settings):
    # generate_jackalsimulator_solver.py:34-56
    modules = ModuleManager()
    model = ContouringSecondOrderUnicycleModel()
    base = modules.add_module(MPCBaseModule(settings))
    base.weigh_variable(var_name="a", weight_names="acceleration")
    base.weigh_variable(var_name="w", weight_names="angular_velocity")
    base.weigh_variable(var_name="v", weight_names=["velocity", "reference_velocity"],
                        cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    modules.add_module(ContouringModule(settings))
    return model, modules


def configuration_basic(settings):
    # generate_jackalsimulator_solver.py:59-64  (cfg 1: MPCC + ellipsoids)
    model, modules = configuration_no_obstacles(settings)
    modules.add_module(EllipsoidConstraintModule(settings))
    return model, modules


def configuration_tmpc(settings):
    # generate_jackalsimulator_solver.py:92-101 (cfg 2/4: T-MPC)
    model, modules = configuration_no_obstacles(settings)
    modules.add_module(GuidanceConstraintModule(settings, constraint_submodule=EllipsoidConstraintModule))
    return model, modules


def scalarize(e):
    if isinstance(e, sp.MatrixBase):
        assert e.shape == (1, 1)
        return e[0, 0]
    return sp.sympify(e)


def num(e, subs):
    return float(sp.N(e.subs(subs), PREC)) if hasattr(e, "subs") else float(e)


def grad_hess(expr, zs, subs):
    g = [sp.diff(expr, s) for s in zs]
    H = [[sp.diff(gi, s) for s in zs] for gi in g]
    return ([num(x, subs) for x in g], [[num(x, subs) for x in row] for row in H])


def make_params_cfg(params, M, S, rng, lin_rows, variant):
    """Numeric parameter vector (index map = reference Parameters order)."""
    p = np.zeros(params.length())
    setp = lambda name, v: p.__setitem__(params._params[name], v)
    # weights: mpc_planner_jackalsimulator/config/settings.yaml:75-89
    for k, v in dict(acceleration=0.34, angular_velocity=0.85, velocity=0.55, reference_velocity=2.0,
                     contour=0.05, lag=0.75, terminal_angle=100.0, terminal_contouring=10.0).items():
        setp(k, v)
    amp = rng.uniform(0.2, 1.0)
    for i in range(S):
        # cubic segments of 6 m: x ~ arc length, y lateral wiggle
        setp(f"spline_x{i}_a", rng.uniform(-2e-3, 2e-3)); setp(f"spline_x{i}_b", rng.uniform(-1e-2, 1e-2))
        setp(f"spline_x{i}_c", 1.0 + rng.uniform(-0.05, 0.05)); setp(f"spline_x{i}_d", 6.0 * i)
        setp(f"spline_y{i}_a", amp * rng.uniform(-5e-3, 5e-3)); setp(f"spline_y{i}_b", amp * rng.uniform(-3e-2, 3e-2))
        setp(f"spline_y{i}_c", amp * rng.uniform(-0.2, 0.2)); setp(f"spline_y{i}_d", amp * rng.uniform(-0.5, 0.5))
        setp(f"spline{i}_start", 6.0 * i)
    setp("ego_disc_radius", 0.325)
    setp("ego_disc_0_offset", 0.0 if variant % 2 == 0 else 0.12)  # exercise the psi-dependence too
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
    if lin_rows:
        g = np.array([rng.uniform(0.0, 6.0), rng.uniform(-1.0, 1.0)])
        for j in range(M):
            o = np.array(obs[j]); a = (o - g) / np.linalg.norm(o - g)
            setp(f"lin_constraint_{j}_a1", a[0]); setp(f"lin_constraint_{j}_a2", a[1])
            setp(f"lin_constraint_{j}_b", a @ o - (1e-3 + 0.325))  # linearized_constraints.cpp:90-105
    return p


def erk4(model, x, u, dt, num_steps):
    """acados ERK: 4 stages (classic RK4 tableau), num_steps sub-steps (generate_acados_solver.py:148-150)
    applied to the reference's continuous_model (solver_model.py:207-214)."""
    h = sp.Rational(1, num_steps) * dt
    f = lambda xx: list(model.continuous_model(xx, u))
    for _ in range(num_steps):
        k1 = f(x)
        k2 = f([xi + h / 2 * ki for xi, ki in zip(x, k1)])
        k3 = f([xi + h / 2 * ki for xi, ki in zip(x, k2)])
        k4 = f([xi + h * ki for xi, ki in zip(x, k3)])
        x = [xi + h / 6 * (a + 2 * b + 2 * c + d) for xi, a, b, c, d in zip(x, k1, k2, k3, k4)]
    return x


def main():
    out = {"_doc": "golden vectors produced by tests/golden/make_golden.py executing the reference's "
                   "python modules under a sympy-backed casadi stand-in; z=[a,w,x,y,psi,v,spline]",
           "cases": []}
    zs = [sp.Symbol(f"z{i}", real=True) for i in range(7)]
    cfgs = [
        ("cfg1_basic_M4", configuration_basic, 4, False),
        ("cfg2_tmpc_M8", configuration_tmpc, 8, True),
        ("cfg4_tmpc_M12", configuration_tmpc, 12, True),
    ]
    for name, conf, M, lin_rows in cfgs:
        settings = base_settings(20, M)
        model, modules = conf(settings)
        params = Parameters()
        define_parameters(modules, params, settings)
        settings["params"] = params
        npar = params.length()
        nh = constraint_number(modules)
        lb = [(-1e15 if v == -np.inf else v) for v in constraint_lower_bounds(modules)]
        ub = [(1e15 if v == np.inf else v) for v in constraint_upper_bounds(modules)]
        pmap = dict(params._params)
        npoints = 4 if M == 8 else 2
        for variant in range(npoints):
            rng = np.random.default_rng(7000 + 100 * M + variant)
            p = make_params_cfg(params, M, 5, rng, lin_rows, variant)
            # z points: variant 1 sits 0.03 m before a knot so the sigmoid glue is exercised
            s_val = [2.6, 5.97, 13.4, 24.2][variant % 4]
            zval = [rng.uniform(-1.5, 1.5), rng.uniform(-0.7, 0.7), s_val + rng.uniform(-0.5, 0.5),
                    rng.uniform(-1.0, 1.0), rng.uniform(-0.6, 0.6), rng.uniform(0.3, 2.5), s_val]
            subs = {zs[i]: sp.Float(repr(zval[i]), PREC) for i in range(7)}
            pl = [sp.Float(repr(float(v)), PREC) for v in p]

            cost = scalarize(objective(modules, zs, pl, model, settings, 1))
            cost_terminal = scalarize(objective(modules, zs, pl, model, settings, settings["N"] - 1))
            cg, cH = grad_hess(cost, zs, subs)
            hs = [scalarize(c) for c in constraints(modules, zs, pl, model, settings, 1)]
            assert len(hs) == nh
            hval, hjac, hhess = [], [], []
            for hexpr in hs:
                hval.append(num(hexpr, subs))
                g, H = grad_hess(hexpr, zs, subs)
                hjac.append(g); hhess.append(H)
            f = list(model.continuous_model(zs[2:], zs[:2]))
            xn = erk4(model, zs[2:], zs[:2], sp.Float("0.2", PREC), 3)
            dval, djac, dhess = [], [], []
            for e in xn:
                dval.append(num(e, subs))
                g, H = grad_hess(e, zs, subs)
                djac.append(g); dhess.append(H)
            out["cases"].append({
                "name": f"{name}_v{variant}", "config": name, "M": M, "S": 5, "npar": npar, "nh": nh,
                "uses_lin_rows": lin_rows, "parameter_map": pmap, "lh": lb, "uh": ub,
                "z": zval, "p": [float(v) for v in p],
                "cost": num(cost, subs), "cost_grad": cg, "cost_hess": cH,
                "cost_forces_terminal_stage": num(cost_terminal, subs),
                "h": hval, "h_jac": hjac, "h_hess": hhess,
                "f_cont": [num(e, subs) for e in f],
                "x_next": dval, "x_next_jac": djac, "x_next_hess": dhess,
            })
            print(name, variant, "cost", out["cases"][-1]["cost"], "npar", npar, "nh", nh, flush=True)

    # Appendix-E style anchors from the reference's own tests (test_control_modules.py:56-59, 89-95)
    settings = {"contouring": {"num_segments": 10, "dynamic_velocity_reference": False}, "N": 20}
    from path_reference_velocity import PathReferenceVelocityModule
    modules = ModuleManager(); modules.add_module(ContouringModule(settings))
    modules.add_module(PathReferenceVelocityModule(settings))
    params = Parameters(); define_parameters(modules, params, settings); settings["params"] = params
    model = ContouringSecondOrderUnicycleModel()
    z = [0.0] * 7; z[3] = 5.0
    obj = objective(modules, [sp.Float(v) for v in z], [sp.Float(1.0)] * params.length(), model, settings, 0)
    out["reference_test_anchors"] = {"test_module_manager_objective": {"npar": params.length(),
                                                                        "objective": float(obj)}}
    settings = {"n_discs": 1, "max_obstacles": 1}
    modules = ModuleManager(); modules.add_module(EllipsoidConstraintModule(settings))
    params = Parameters(); define_parameters(modules, params, settings); settings["params"] = params
    p = [0.0] * params.length(); p[2] = 5.0; p[3] = 10.0; p[-1] = 1.0
    c = constraints(modules, [sp.Float(0.0)] * 7, [sp.Float(v) for v in p], model, settings, 0)
    out["reference_test_anchors"]["test_module_manager_constraints"] = {
        "npar": params.length(), "constraint": [float(scalarize(x)) for x in c]}
    print(out["reference_test_anchors"])

    with open(os.path.join(HERE, "stage_functions.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", os.path.join(HERE, "stage_functions.json"))


if __name__ == "__main__":
    main()
