"""Input of tools/scan_vs_riccati_bench.hip: 64 REAL interior-point Newton systems of the cfg-2 QP (N = 20, nu = 2, nx = 5) -- the
oracle's debug dump of an RTI iteration of bench scenes, put on interior-point iterates of decreasing barrier parameter exactly as
tools/parallel_in_time_study.py does -- with the solution of each from a dense KKT solve refined in extended precision.

    python tools/make_newton_systems.py            ->  build/newton_systems.bin   (git-ignored; travels to the GPU box)

Layout (little endian): int32 B, N, NV, NX ; then per system float64 Hh[(N+1)*49], gh[(N+1)*7], BA[N*35], rb[N*5], dz[(N+1)*7].
Stage 0 uses its input block only (dx_0 = 0), stage N its state block only (as in the kernels)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O
from mpc_planner_amd import scenes
import parallel_in_time_study as S

NV, NX, NU = S.NV, S.NX, S.NU


def solve_refined(Hh, gh, BA, rb, N):
    """Dense KKT (float64 LU) + iterative refinement with the residual in long double: the reference solution."""
    idx_free = [k * NV + i for k in range(N + 1) for i in range(NV) if not (k == 0 and i >= NU) and not (k == N and i < NU)]
    nf = len(idx_free); pos = {e: i for i, e in enumerate(idx_free)}
    n = nf + N * NX
    K = np.zeros((n, n)); r = np.zeros(n)
    for k in range(N + 1):
        for i in range(NV):
            if k * NV + i not in pos: continue
            a = pos[k * NV + i]; r[a] = -gh[k, i]
            for j in range(NV):
                if k * NV + j in pos: K[a, pos[k * NV + j]] = Hh[k, i, j]
    for k in range(N):
        for m in range(NX):
            row = nf + k * NX + m
            for j in range(NV):
                if k * NV + j in pos:
                    K[row, pos[k * NV + j]] = BA[k, m, j]; K[pos[k * NV + j], row] = BA[k, m, j]
            c = pos[(k + 1) * NV + NU + m]
            K[row, c] = -1; K[c, row] = -1
            r[row] = -rb[k, m]
    import scipy.linalg as sl
    lu = sl.lu_factor(K)
    x = sl.lu_solve(lu, r)
    Kl, rl = K.astype(np.longdouble), r.astype(np.longdouble)
    for _ in range(4):
        res = (rl - Kl @ x.astype(np.longdouble)).astype(np.float64)
        x = x + sl.lu_solve(lu, res)
    dz = np.zeros((N + 1, NV))
    for e, a in pos.items():
        dz[e // NV, e % NV] = x[a]
    return dz


def main():
    N, M, B = 20, 8, 64
    pb = O.problem(N=N, S=5, n_lin=M, M=M)
    lb = np.array([-2.0, -0.8, -2000.0, -2000.0, -4 * np.pi, -0.01, -1.0]); ub = np.array([2.0, 0.8, 2000.0, 2000.0, 4 * np.pi, 3.0, 10000.0])
    rng = np.random.default_rng(0)
    out = []
    mus = (1e-1, 1e-2, 1e-4, 1e-6)
    worst = 0.0
    for i in range(B // len(mus)):
        scene, b_, it = 3 + i, (7 * i + 5) % 64, 1 + i % 8
        sc = scenes.make_scene(scene, N=N, M=M, B=64)
        _, _, _, dbg = O.solve(pb, sc["xinit"][b_], sc["x0"][b_].reshape(-1), sc["params"][b_].reshape(-1), debug_iter=it)
        z = np.array(dbg.z_in[:(N + 1) * 7]).reshape(N + 1, 7)
        for mu in mus:
            Hh, gh, BA, rb, _, _ = S.newton_system(dbg, N, 2 * M, M, mu, rng, lb, ub, z)
            dz = solve_refined(Hh, gh, BA, rb, N)
            if mu == 1e-4 and i < 3:                                            # cross-check the reference against the 40-digit solve
                ex = S.solve_exact(Hh, gh, BA, rb, N)
                worst = max(worst, float((np.abs(dz - ex) / np.maximum(np.abs(ex).max(axis=1, keepdims=True), 1e-6)).max()))
            out.append(np.concatenate([Hh.ravel(), gh.ravel(), BA.ravel(), rb.ravel(), dz.ravel()]))
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    path = os.path.join(ROOT, "build", "newton_systems.bin")
    with open(path, "wb") as fh:
        fh.write(np.array([B, N, NV, NX], np.int32).tobytes())
        fh.write(np.stack(out).astype("<f8").tobytes())
    print(path, len(out), "systems; refined reference vs 40-digit solve:", worst)


if __name__ == "__main__":
    main()
