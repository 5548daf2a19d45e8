"""Verdict round 2, next-2 ("measure a parallel-in-time Riccati"): the numerics question that comes before any kernel.

Every parallel-in-time solve of the interior-point Newton system (block cyclic reduction, an associative scan over the stages)
works on a REDUCED system -- here the block-tridiagonal Schur complement in the dynamics multipliers,
    Y pi = beta,   Y_kk = [B A]_k H_k^-1 [B A]_k^T + (H_{k+1}^-1)_xx,   Y_{k,k+1} = -([B A]_{k+1} H_{k+1}^-1)_{.,x}^T ...
whose blocks need H_k^-1 of the barrier-augmented stage Hessians -- while the production kernels use the square-root Riccati
recursion on the stage blocks themselves.  This script takes REAL QPs of the bench scenes (the oracle's debug dump of an RTI
iteration: W, g, [B A], b, rows), puts them on interior-point iterates of decreasing barrier parameter mu (active rows get
lambda / t up to 1e12, inactive ones down to 1e-10: what late iterations look like), and solves the same Newton system three ways:
    exact      dense KKT solve in 40-digit arithmetic (mpmath)
    riccati    the sequential square-root recursion in float64 (the kernels' algorithm, numpy restatement)
    schur      the multiplier Schur complement in float64, solved by block cyclic reduction over the stages (the parallel form)
and reports the relative error of the primal step per stage against the north-star tolerance 1e-4 (and the 2e-5 the tests assert).
Writes profiles/round3_d_parallel_in_time_numerics.json.   CPU only:  python tools/parallel_in_time_study.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpmath as mp
import oracle_lib as O
from mpc_planner_amd import scenes

NV, NX, NU = 7, 5, 2
mp.mp.dps = 40


def newton_system(dbg, N, nh, n_up, mu, rng, lb, ub, z):
    """Stage blocks Hh_k, gh_k, [B A]_k, rb_k of an interior-point Newton system at barrier parameter mu."""
    W = np.array(dbg.W[:(N + 1) * 49]).reshape(N + 1, 7, 7); g = np.array(dbg.g[:(N + 1) * 7]).reshape(N + 1, 7)
    BA = np.array(dbg.BA[:N * 35]).reshape(N, 5, 7); b = np.array(dbg.b[:N * 5]).reshape(N, 5)
    D = np.array(dbg.D[:N * O.MAX_NH * 7]).reshape(N, O.MAX_NH, 7); h = np.array(dbg.h[:N * O.MAX_NH]).reshape(N, O.MAX_NH)
    dz = np.array(dbg.dz[:(N + 1) * 7]).reshape(N + 1, 7)
    Hh = W.copy(); gh = g.copy(); dmax, dmin = 0.0, np.inf
    for k in range(N):
        rows = []
        for r in range(nh):                                                       # general rows: signed so that  c.dz <= beta
            sgn = 1.0 if r < n_up else -1.0
            bound = 0.0 if r < n_up else 1.0
            rows.append((sgn * D[k, r], sgn * (bound - h[k, r])))
        for i in range(NV if k >= 1 else NU):                                     # boxes (x_0 is fixed)
            e = np.zeros(7); e[i] = 1.0
            rows.append((e, ub[i] - z[k, i])); rows.append((-e, -(lb[i] - z[k, i])))
        for c, beta in rows:
            t_star = beta - c @ dz[k]
            lam_star = rng.uniform(1.0, 100.0)                                    # multiplier of an active row at the solution
            t = max(t_star, mu / lam_star); lam = mu / t                          # on the central path: lam t = mu
            d = lam / t
            dmax, dmin = max(dmax, d), min(dmin, d)
            Hh[k] += d * np.outer(c, c)
            gh[k] += c * (lam + d * (beta - c @ dz[k] - t))
    return Hh, gh, BA, b, dmax, dmin


def solve_exact(Hh, gh, BA, rb, N):
    """Dense KKT in 40-digit arithmetic: variables dz_0..dz_N (dx_0 = 0 eliminated by huge weight is avoided: rows removed), pi_1..pi_N."""
    nz = (N + 1) * NV
    idx_free = [k * NV + i for k in range(N + 1) for i in range(NV) if not (k == 0 and i >= NU) and not (k == N and i < NU)]
    nf = len(idx_free); pos = {e: i for i, e in enumerate(idx_free)}
    n = nf + N * NX
    K = mp.zeros(n, n); r = mp.zeros(n, 1)
    for k in range(N + 1):
        for i in range(NV):
            if k * NV + i not in pos: continue
            a = pos[k * NV + i]
            r[a] = -mp.mpf(float(gh[k, i]))
            for j in range(NV):
                if k * NV + j in pos: K[a, pos[k * NV + j]] = mp.mpf(float(Hh[k, i, j]))
    for k in range(N):                                                             # dx_{k+1} = BA_k dz_k + rb_k
        for m in range(NX):
            row = nf + k * NX + m
            for j in range(NV):
                if k * NV + j in pos:
                    K[row, pos[k * NV + j]] = mp.mpf(float(BA[k, m, j])); K[pos[k * NV + j], row] = mp.mpf(float(BA[k, m, j]))
            c = pos[(k + 1) * NV + NU + m]
            K[row, c] = -1; K[c, row] = -1
            r[row] = -mp.mpf(float(rb[k, m]))
    sol = mp.lu_solve(K, r)
    dz = np.zeros((N + 1, NV))
    for e, a in pos.items():
        dz[e // NV, e % NV] = float(sol[a])
    return dz


def solve_riccati(Hh, gh, BA, rb, N):
    """Square-root Riccati recursion in float64 (tmpc_riccati.hpp restated): F = Hh + G^T G, G = Lp^T [B A]."""
    Lp = np.linalg.cholesky(Hh[N][NU:, NU:]); p = gh[N][NU:].copy()
    fac = [None] * N
    for k in range(N - 1, -1, -1):
        G = Lp.T @ BA[k]
        F = Hh[k] + G.T @ G
        f = gh[k] + BA[k].T @ (Lp @ (Lp.T @ rb[k]) + p)
        if k == 0:                                                                 # dx_0 = 0: only the input block
            Luu = np.linalg.cholesky(F[:NU, :NU]); fac[0] = (Luu, None, f); break
        L = np.linalg.cholesky(F)
        Luu, Lxu, Lxx = L[:NU, :NU], L[NU:, :NU], L[NU:, NU:]
        y = np.linalg.solve(Luu, f[:NU])
        p = f[NU:] - Lxu @ y
        fac[k] = (Luu, Lxu, y)
        Lp = Lxx
    dz = np.zeros((N + 1, NV)); dx = np.zeros(NX)
    for k in range(N):
        if k == 0:
            Luu, _, f = fac[0]; du = -np.linalg.solve(Luu.T, np.linalg.solve(Luu, f[:NU]))
        else:
            Luu, Lxu, y = fac[k]; du = -np.linalg.solve(Luu.T, Lxu.T @ dx + y)
        dz[k, :NU] = du; dz[k, NU:] = dx
        dx = BA[k] @ dz[k] + rb[k]
    dz[N, NU:] = dx
    return dz


def solve_schur_cr(Hh, gh, BA, rb, N):
    """Multiplier Schur complement, float64, block cyclic reduction over the stages (what a parallel-in-time kernel would run).
    Stage k variables dz_k (stage 0: inputs only, stage N: states only); constraints k = 0..N-1:  C_k dz_k - E dz_{k+1} = -rb_k."""
    Hs, gs, Cs = [], [], []
    for k in range(N + 1):
        sel = list(range(NU)) if k == 0 else (list(range(NU, NV)) if k == N else list(range(NV)))
        Hs.append(Hh[k][np.ix_(sel, sel)]); gs.append(gh[k][sel])
        Cs.append(BA[k][:, sel] if k < N else None)
    Es = [None] + [(-np.eye(NV)[NU:, :][:, list(range(NU, NV)) if k == N else list(range(NV))]) for k in range(1, N + 1)]   # coefficient of dz_k in constraint k-1
    Hinv = [np.linalg.inv(Hk) for Hk in Hs]
    # Y pi = beta, block tridiagonal: diag[k] = C_k Hinv_k C_k^T + E_{k+1} Hinv_{k+1} E_{k+1}^T ; off[k] = E_{k+1} Hinv_{k+1} C_{k+1}^T (couples k, k+1)
    diag = [Cs[k] @ Hinv[k] @ Cs[k].T + Es[k + 1] @ Hinv[k + 1] @ Es[k + 1].T for k in range(N)]
    off = [Es[k + 1] @ Hinv[k + 1] @ Cs[k + 1].T for k in range(N - 1)]
    beta = [rb[k] - Cs[k] @ Hinv[k] @ gs[k] - Es[k + 1] @ Hinv[k + 1] @ gs[k + 1] for k in range(N)]

    def cr(diag, off, beta):
        n = len(diag)
        if n == 1:
            return [np.linalg.solve(diag[0], beta[0])]
        ev = list(range(0, n, 2)); od = list(range(1, n, 2))
        # eliminate the odd blocks: pi_o = D_o^-1 (beta_o - off[o-1]^T... )
        Dinv = {o: np.linalg.inv(diag[o]) for o in od}
        nd, no, nb = [], [], []
        for i, e in enumerate(ev):
            d = diag[e].copy(); bb = beta[e].copy()
            if e - 1 >= 0:
                L = off[e - 1].T            # couples (e-1, e): row e, col e-1 block = off[e-1]^T
                d -= L @ Dinv[e - 1] @ L.T; bb -= L @ Dinv[e - 1] @ beta[e - 1]
            if e + 1 < n:
                U = off[e]                  # row e, col e+1
                d -= U @ Dinv[e + 1] @ U.T; bb -= U @ Dinv[e + 1] @ beta[e + 1]
            nd.append(d); nb.append(bb)
            if e + 2 < n:
                no.append(-off[e] @ Dinv[e + 1] @ off[e + 1])
        xe = cr(nd, no, nb)
        x = [None] * n
        for i, e in enumerate(ev): x[e] = xe[i]
        for o in od:
            rhs = beta[o] - off[o - 1].T @ x[o - 1]
            if o + 1 < n: rhs = rhs - off[o] @ x[o + 1]
            x[o] = Dinv[o] @ rhs
        return x

    pi = cr(diag, off, beta)
    dz = np.zeros((N + 1, NV))
    for k in range(N + 1):
        rhs = -gs[k].copy()                                         # H dz = -(g + C^T pi_k + E^T pi_{k-1})
        if k < N: rhs = rhs - Cs[k].T @ pi[k]
        if k >= 1: rhs = rhs - Es[k].T @ pi[k - 1]
        v = Hinv[k] @ rhs
        sel = list(range(NU)) if k == 0 else (list(range(NU, NV)) if k == N else list(range(NV)))
        dz[k, sel] = v
    return dz, max(np.linalg.cond(d) for d in diag)


def main():
    N, M = 20, 8
    pb = O.problem(N=N, S=5, n_lin=M, M=M)
    lb = np.array([-2.0, -0.8, -2000.0, -2000.0, -4 * np.pi, -0.01, -1.0]); ub = np.array([2.0, 0.8, 2000.0, 2000.0, 4 * np.pi, 3.0, 10000.0])
    rng = np.random.default_rng(0)
    out = []
    for scene, b_, it in ((3, 5, 2), (3, 40, 6), (11, 17, 4)):
        sc = scenes.make_scene(scene, N=N, M=M, B=64)
        xt, ut, info, dbg = O.solve(pb, sc["xinit"][b_], sc["x0"][b_].reshape(-1), sc["params"][b_].reshape(-1), debug_iter=it)
        z = np.array(dbg.z_in[:(N + 1) * 7]).reshape(N + 1, 7)
        for mu in (1e-2, 1e-4, 1e-6, 1e-8):
            Hh, gh, BA, rb, dmax, dmin = newton_system(dbg, N, 2 * M, M, mu, rng, lb, ub, z)
            ex = solve_exact(Hh, gh, BA, rb, N)
            ri = solve_riccati(Hh, gh, BA, rb, N)
            sr, condY = solve_schur_cr(Hh, gh, BA, rb, N)
            scale = np.maximum(np.abs(ex).max(axis=1, keepdims=True), 1e-6)
            rec = dict(scene=scene, trajectory=b_, rti_iteration=it, mu=mu, barrier_weight_range=[dmin, dmax],
                       riccati_max_rel_err=float((np.abs(ri - ex) / scale).max()), schur_cyclic_reduction_max_rel_err=float((np.abs(sr - ex) / scale).max()),
                       schur_block_condition_max=float(condY))
            print(json.dumps(rec), flush=True)
            out.append(rec)
    worst_r = max(r["riccati_max_rel_err"] for r in out); worst_s = max(r["schur_cyclic_reduction_max_rel_err"] for r in out)
    summary = dict(what=__doc__.split("\n\n")[0], systems=out, worst_riccati=worst_r, worst_schur_cyclic_reduction=worst_s,
                   north_star_tolerance=1e-4, tests_assert=2e-5)
    with open(os.path.join(ROOT, "profiles", "round3_d_parallel_in_time_numerics.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("worst riccati", worst_r, "worst schur+CR", worst_s)


if __name__ == "__main__":
    main()
