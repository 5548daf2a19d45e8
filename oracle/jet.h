/*
 * oracle/jet.h -- CPU ORACLE (test infrastructure): forward-mode second-order jets in ORC_NVE variables.
 *
 * A jet carries (value, gradient, Hessian) of a scalar w.r.t. z = [a,w,x,y,psi,v,spline(,slack)].  Composing
 * jets through the reference's expressions reproduces what CasADi's AD yields for the same expression
 * graph (generate_acados_solver.py:41,48 build the expressions; acados' EXACT Hessian uses their
 * second derivatives).
 */
#ifndef ORC_JET_H
#define ORC_JET_H

#include <math.h>
#include <string.h>
#include "tmpc_oracle.h"

typedef struct {
    double v;
    double g[ORC_NVE];
    double H[ORC_NVE][ORC_NVE];
} jet;

static inline jet jet_const(double c) { jet r; memset(&r, 0, sizeof r); r.v = c; return r; }
static inline jet jet_var(double val, int idx) { jet r = jet_const(val); r.g[idx] = 1.0; return r; }

/* r = f(a) given f, f', f'' at a.v */
static inline jet jet_chain(const jet *a, double f, double f1, double f2) {
    jet r; r.v = f;
    for (int i = 0; i < ORC_NVE; i++) r.g[i] = f1 * a->g[i];
    for (int i = 0; i < ORC_NVE; i++)
        for (int j = 0; j < ORC_NVE; j++)
            r.H[i][j] = f1 * a->H[i][j] + f2 * a->g[i] * a->g[j];
    return r;
}
static inline jet jet_add(jet a, jet b) {
    jet r; r.v = a.v + b.v;
    for (int i = 0; i < ORC_NVE; i++) r.g[i] = a.g[i] + b.g[i];
    for (int i = 0; i < ORC_NVE; i++) for (int j = 0; j < ORC_NVE; j++) r.H[i][j] = a.H[i][j] + b.H[i][j];
    return r;
}
static inline jet jet_sub(jet a, jet b) {
    jet r; r.v = a.v - b.v;
    for (int i = 0; i < ORC_NVE; i++) r.g[i] = a.g[i] - b.g[i];
    for (int i = 0; i < ORC_NVE; i++) for (int j = 0; j < ORC_NVE; j++) r.H[i][j] = a.H[i][j] - b.H[i][j];
    return r;
}
static inline jet jet_mul(jet a, jet b) {
    jet r; r.v = a.v * b.v;
    for (int i = 0; i < ORC_NVE; i++) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
    for (int i = 0; i < ORC_NVE; i++)
        for (int j = 0; j < ORC_NVE; j++)
            r.H[i][j] = a.v * b.H[i][j] + b.v * a.H[i][j] + a.g[i] * b.g[j] + a.g[j] * b.g[i];
    return r;
}
static inline jet jet_scale(jet a, double c) { return jet_chain(&a, c * a.v, c, 0.0); }
static inline jet jet_addc(jet a, double c) { a.v += c; return a; }
static inline jet jet_neg(jet a) { return jet_scale(a, -1.0); }
static inline jet jet_recip(jet a) {
    double r = 1.0 / a.v;
    return jet_chain(&a, r, -r * r, 2.0 * r * r * r);
}
static inline jet jet_div(jet a, jet b) { return jet_mul(a, jet_recip(b)); }
static inline jet jet_sqrt(jet a) {
    double s = sqrt(a.v);
    return jet_chain(&a, s, 0.5 / s, -0.25 / (s * a.v));
}
static inline jet jet_exp(jet a) { double e = exp(a.v); return jet_chain(&a, e, e, e); }
static inline jet jet_sin(jet a) { double s = sin(a.v), c = cos(a.v); return jet_chain(&a, s, c, -s); }
static inline jet jet_cos(jet a) { double s = sin(a.v), c = cos(a.v); return jet_chain(&a, c, -s, -c); }
static inline jet jet_sq(jet a) { return jet_mul(a, a); }

#endif
