"""Sparse forward-mode second-order differentiation at code level.

emit.py's first version differentiated the stage functions symbolically and let sympy's CSE share subexpressions; the
result is correct but long-lived: hundreds of temporaries that all stay live until the 36 outputs are formed.  This module
walks the expression DAG once and, for every node, emits its value together with only those first and second partials
that can be non-zero -- a node that depends on the spline coordinate alone carries (v, d/ds, d2/ds2), exactly what the
hand-written cost does with its one-variable Taylor triples (csrc/tmpc_stage.hpp) -- so partials are consumed right where
they are produced.

A node's derivative information is kept symbolically as C expressions (strings) or python floats; structural zeros are
never emitted, constant partials (d x / d x = 1) are folded.
"""
import sympy as sp


class _Jet:
    """value: C expression; g: {i: expr}; h: {(i, j) with i >= j: expr}.  expr is a str (C code) or a float (constant)."""
    __slots__ = ("v", "g", "h")

    def __init__(self, v, g=None, h=None):
        self.v, self.g, self.h = v, g or {}, h or {}


def _is0(x):
    return isinstance(x, float) and x == 0.0


def _mul(a, b):
    if _is0(a) or _is0(b):
        return 0.0
    if isinstance(a, float) and isinstance(b, float):
        return a * b
    if isinstance(a, float):
        a, b = b, a
    if isinstance(b, float):
        if b == 1.0:
            return a
        if b == -1.0:
            return f"(-{a})"
        return f"({b!r}*{a})"
    return f"({a}*{b})"


def _add(a, b):
    if _is0(a):
        return b
    if _is0(b):
        return a
    if isinstance(a, float) and isinstance(b, float):
        return a + b
    a_ = repr(a) if isinstance(a, float) else a
    b_ = repr(b) if isinstance(b, float) else b
    return f"({a_} + {b_})"


def _sum(terms):
    out = 0.0
    for t in terms:
        out = _add(out, t)
    return out


class JetEmitter:
    """Turns sympy expressions into C statements computing value / gradient / Hessian w.r.t. `variables` (sympy symbols),
    restricted to `wrt` (indices into variables) -- everything else is a plain scalar."""

    def __init__(self, variables, printer, prefix):
        self.vars = {v: i for i, v in enumerate(variables)}
        self.c = printer
        self.prefix = prefix
        self.lines = []
        self.cache = {}
        self.n = 0

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _tmp(self, expr):
        """Materialise a C expression in a temporary (floats and plain identifiers stay as they are)."""
        if isinstance(expr, float):
            return expr
        if expr.replace("_", "").isalnum() and not expr[0].isdigit():
            return expr
        name = f"{self.prefix}{self.n}"; self.n += 1
        self.lines.append(f"    const double {name} = {expr};")
        return name

    def _fix(self, jet):
        jet.v = self._tmp(jet.v) if not isinstance(jet.v, float) else jet.v
        jet.g = {i: self._tmp(e) for i, e in jet.g.items() if not _is0(e)}
        jet.h = {k: self._tmp(e) for k, e in jet.h.items() if not _is0(e)}
        return jet

    @staticmethod
    def _H(j, i, k):
        return j.h.get((i, k) if i >= k else (k, i), 0.0)

    def _unary(self, u, f, f1, f2):
        """r = f(u) with f' = f1, f'' = f2 (C expressions / floats)."""
        f1 = self._tmp(f1) if u.g else f1
        f2 = self._tmp(f2) if (u.g and not _is0(f2)) else f2
        g = {i: _mul(f1, e) for i, e in u.g.items()}
        h = {}
        idx = sorted(u.g)
        for a, i in enumerate(idx):
            for k in idx[:a + 1]:
                h[(i, k)] = _add(_mul(f1, self._H(u, i, k)), _mul(f2, _mul(u.g[i], u.g[k])))
        return self._fix(_Jet(f, g, h))

    def _binary_mul(self, a, b):
        if not a.g and not b.g:
            return self._fix(_Jet(_mul(a.v, b.v)))
        idx = sorted(set(a.g) | set(b.g))
        g = {i: _add(_mul(a.v, b.g.get(i, 0.0)), _mul(b.v, a.g.get(i, 0.0))) for i in idx}
        h = {}
        for p, i in enumerate(idx):
            for k in idx[:p + 1]:
                t = _add(_mul(a.v, self._H(b, i, k)), _mul(b.v, self._H(a, i, k)))
                t = _add(t, _mul(a.g.get(i, 0.0), b.g.get(k, 0.0)))
                t = _add(t, _mul(a.g.get(k, 0.0), b.g.get(i, 0.0)))
                h[(i, k)] = t
        return self._fix(_Jet(_mul(a.v, b.v), g, h))

    # ---- the DAG walk ------------------------------------------------------------------------------------------
    def jet(self, e):
        e = sp.sympify(e)
        if e in self.cache:
            return self.cache[e]
        r = self._build(e)
        self.cache[e] = r
        return r

    def _build(self, e):
        if e.is_Number or e.is_NumberSymbol:
            return _Jet(float(e))
        if e.is_Symbol:
            if e in self.vars:
                return _Jet(self.c(e), {self.vars[e]: 1.0})
            return _Jet(self.c(e))
        if not (e.free_symbols & self.vars.keys()):                 # parameter-only subtree: one scalar expression
            return self._fix(_Jet(self.c(e)))
        if e.is_Add:
            js = [self.jet(a) for a in e.args]
            idx = sorted(set().union(*[set(j.g) for j in js]))
            g = {i: _sum(j.g.get(i, 0.0) for j in js) for i in idx}
            h = {}
            for p, i in enumerate(idx):
                for k in idx[:p + 1]:
                    h[(i, k)] = _sum(self._H(j, i, k) for j in js)
            return self._fix(_Jet(_sum(j.v for j in js), g, h))
        if e.is_Mul:
            js = [self.jet(a) for a in e.args]
            js.sort(key=lambda j: len(j.g))                          # scalars first: their product stays a scalar
            r = js[0]
            for j in js[1:]:
                r = self._binary_mul(r, j)
            return r
        if e.is_Pow:
            b, x = e.as_base_exp()
            if not x.is_Number:
                return self.jet(sp.exp(x * sp.log(b)))
            u = self.jet(b)
            uv = u.v
            if x == 2:
                return self._unary(u, _mul(uv, uv), _mul(2.0, uv), 2.0)
            if x == -1:
                f = self._tmp(f"(1.0/{uv})")
                return self._unary(u, f, f"(-{f}*{f})", f"(2.0*{f}*{f}*{f})")
            if x == sp.Rational(1, 2):
                f = self._tmp(f"sqrt({uv})")
                return self._unary(u, f, f"(0.5/{f})", f"(-0.25/({f}*{uv}))")
            if x == -sp.Rational(1, 2):
                f = self._tmp(f"(1.0/sqrt({uv}))")
                return self._unary(u, f, f"(-0.5*{f}/{uv})", f"(0.75*{f}/({uv}*{uv}))")
            if x.is_Integer and 3 <= int(x) <= 4:
                n = int(x)
                um2 = _mul(uv, uv) if n == 4 else uv                  # u^(n-2)
                um2 = self._tmp(um2)
                um1 = self._tmp(_mul(um2, uv))
                return self._unary(u, _mul(um1, uv), _mul(float(n), um1), _mul(float(n * (n - 1)), um2))
            c = float(x)
            f = self._tmp(f"pow({uv}, {c!r})")
            return self._unary(u, f, f"({c!r}*{f}/{uv})", f"({c * (c - 1.0)!r}*{f}/({uv}*{uv}))")
        name = e.func.__name__
        if name == "atan2":
            y, x = self.jet(e.args[0]), self.jet(e.args[1])
            d = self._tmp(f"({x.v}*{x.v} + {y.v}*{y.v})")
            fy, fx = self._tmp(f"({x.v}/{d})"), self._tmp(f"(-{y.v}/{d})")
            fxy = self._tmp(f"(({y.v}*{y.v} - {x.v}*{x.v})/({d}*{d}))")
            fyy = self._tmp(f"(-2.0*{x.v}*{y.v}/({d}*{d}))")
            idx = sorted(set(x.g) | set(y.g))
            g = {i: _add(_mul(fy, y.g.get(i, 0.0)), _mul(fx, x.g.get(i, 0.0))) for i in idx}
            h = {}
            for p, i in enumerate(idx):
                for k in idx[:p + 1]:
                    yi, yk, xi, xk = y.g.get(i, 0.0), y.g.get(k, 0.0), x.g.get(i, 0.0), x.g.get(k, 0.0)
                    t = _add(_mul(fy, self._H(y, i, k)), _mul(fx, self._H(x, i, k)))
                    t = _add(t, _mul(fyy, _add(_mul(yi, yk), _mul(-1.0, _mul(xi, xk)))))      # f_xx = -f_yy
                    t = _add(t, _mul(fxy, _add(_mul(yi, xk), _mul(yk, xi))))
                    h[(i, k)] = t
            return self._fix(_Jet(f"atan2({y.v}, {x.v})", g, h))
        if name in ("Max", "Min"):
            js = [self.jet(a) for a in e.args]
            r = js[0]
            for j in js[1:]:
                cond = self._tmp(f"(double)({r.v} {'>=' if name == 'Max' else '<='} {j.v})")     # 1.0 / 0.0
                pick = lambda a, b: _add(_mul(cond, a), _mul(f"(1.0 - {cond})", b))
                idx = sorted(set(r.g) | set(j.g))
                g = {i: pick(r.g.get(i, 0.0), j.g.get(i, 0.0)) for i in idx}
                h = {(i, k): pick(self._H(r, i, k), self._H(j, i, k)) for p, i in enumerate(idx) for k in idx[:p + 1]}
                r = self._fix(_Jet(f"f{name.lower()}({r.v}, {j.v})", g, h))
            return r
        u = self.jet(e.args[0])
        uv = u.v
        if name == "exp":
            f = self._tmp(f"exp({uv})"); return self._unary(u, f, f, f)
        if name == "log":
            return self._unary(u, f"log({uv})", f"(1.0/{uv})", f"(-1.0/({uv}*{uv}))")
        if name == "sin":
            s_, c_ = self._tmp(f"sin({uv})"), self._tmp(f"cos({uv})"); return self._unary(u, s_, c_, f"(-{s_})")
        if name == "cos":
            s_, c_ = self._tmp(f"sin({uv})"), self._tmp(f"cos({uv})"); return self._unary(u, c_, f"(-{s_})", f"(-{c_})")
        if name == "tan":
            t = self._tmp(f"tan({uv})"); f1 = self._tmp(f"(1.0 + {t}*{t})"); return self._unary(u, t, f1, f"(2.0*{t}*{f1})")
        if name == "erf":
            f1 = self._tmp(f"(1.1283791670955126*exp(-{uv}*{uv}))"); return self._unary(u, f"erf({uv})", f1, f"(-2.0*{uv}*{f1})")
        if name == "logistic":
            f = self._tmp(f"tmpc_gen_logistic({uv})"); f1 = self._tmp(f"({f}*(1.0 - {f}))")
            return self._unary(u, f, f1, f"({f1}*(1.0 - 2.0*{f}))")
        if name == "Abs":
            return self._unary(u, f"fabs({uv})", f"(({uv} >= 0.0) ? 1.0 : -1.0)", 0.0)
        if name == "atan":
            d = self._tmp(f"(1.0 + {uv}*{uv})"); return self._unary(u, f"atan({uv})", f"(1.0/{d})", f"(-2.0*{uv}/({d}*{d}))")
        raise NotImplementedError(f"no derivative rule for {name}")


def _lit(x):
    return repr(x) if isinstance(x, float) else x
