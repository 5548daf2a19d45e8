"""`casadi`-compatible symbolic facade backed by sympy.

Covers what the reference's module scripts use (grep of `cd.` / `casadi.` over mpc_planner_modules/scripts and
solver_generator: SX, SX.sym, vertcat, cos, sin, tan, atan, atan2, sqrt, exp, log, erf, fmax, fmin, fabs, fmod, pi).
Scalars are sympy expressions; vectors are numpy object arrays, which is also what the module scripts build themselves
(`np.array([x, y])`, `.dot`, `@`).  `install_as_casadi()` registers this module as `casadi` so that unmodified module
scripts `import casadi as cd`.
"""
import sys
import types

import numpy as np
import sympy as sp

# numpy calls obj.exp() / obj.sqrt() / ... when a ufunc meets an object array (spline.py:37 does np.exp(symbol))
for _name, _fn in (("exp", sp.exp), ("sqrt", sp.sqrt), ("cos", sp.cos), ("sin", sp.sin), ("log", sp.log)):
    setattr(sp.Expr, _name, (lambda f: (lambda self: f(self)))(_fn))

pi = sp.pi
cos, sin, tan = sp.cos, sp.sin, sp.tan
atan, arctan, atan2 = sp.atan, sp.atan, sp.atan2
sqrt, exp, log, erf = sp.sqrt, sp.exp, sp.log, sp.erf
fabs = sp.Abs


def fmax(a, b):
    return sp.Max(a, b)


def fmin(a, b):
    return sp.Min(a, b)


def fmod(a, b):
    return a - b * sp.floor(a / b)


def _flatten(items):
    out = []
    for a in items:
        if isinstance(a, sp.MatrixBase):
            out.extend(list(a))
        elif isinstance(a, (list, tuple, np.ndarray)):
            out.extend(_flatten(list(a)))
        else:
            out.append(sp.sympify(a))
    return out


def vertcat(*args):
    return np.array(_flatten(args), dtype=object)


class _SXMeta(type):
    def __call__(cls, *args):
        if len(args) == 2 and all(isinstance(a, int) for a in args):
            return np.zeros(args, dtype=object) + sp.Integer(0)
        if len(args) == 1:
            a = args[0]
            if isinstance(a, (np.ndarray, list, tuple)):
                return np.array(a, dtype=object)
            if isinstance(a, sp.MatrixBase):
                return np.array(a.tolist(), dtype=object)
            return sp.sympify(a)
        if not args:
            return np.zeros((0,), dtype=object)
        raise NotImplementedError(f"SX{args}")


class SX(metaclass=_SXMeta):
    @staticmethod
    def sym(name, n=1, m=1):
        if n == 1 and m == 1:
            return sp.Symbol(name, real=True)
        return np.array([sp.Symbol(f"{name}_{i}", real=True) for i in range(n * m)], dtype=object)


MX = SX


def install_as_casadi():
    """Make `import casadi` resolve to this facade (for unmodified module scripts)."""
    if not hasattr(np, "Inf"):
        np.Inf = np.inf             # module scripts written for NumPy 1.x (gaussian_constraints.py:65) keep working under NumPy 2
    mod = types.ModuleType("casadi")
    for k, v in globals().items():
        if not k.startswith("_") and k not in ("sys", "types", "np", "sp"):
            setattr(mod, k, v)
    sys.modules["casadi"] = mod
    return mod
