"""One control tick (host call -> best index on host) per kernel variant, for shapes bench.py's latency block does not cover -- above all the horizon the
reference SHIPS (mpc_planner_jackalsimulator/config/settings.yaml N: 30, mpc_planner_jackal likewise; guidance_planner.yaml n_paths: 4) -- with the
parity of every variant against the oracle on the same launch.  One JSON line per (shape, planners).
Usage: python tools/tick_shapes.py [reps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from mpc_planner_amd import scenes, solver  # noqa: E402

SHAPES = {
    "jackalsimulator as shipped (settings.yaml: N 30, max_obstacles 12 -> 12 + 12 rows)": (dict(N=30, S=5, n_lin=12, M=12), None, dict(N=30, M=12, tmpc_pp=True)),
    "jackalsimulator stack at the shipped horizon, configs[1]'s 8 obstacles (N 30, 8 + 8 rows)": (dict(N=30, S=5, n_lin=8, M=8), None, dict(N=30, M=8, tmpc_pp=True)),
    "jackal default (N 30, Gaussian rows, 5 obstacles)": (dict(N=30, S=3, n_lin=5, M=5, row_model=1), dict(N=30, S=3, n_lin=5, M=0, n_gauss=5), dict(N=30, M=5, S=3, chance=True)),
    "cfg3 rosnavigation stack (N 30, slack, 8 + 12 + 8 rows)": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), None, dict(N=30, M=8, slack=True, n_decomp=12)),
    "cfg3 as named (curvature-aware cost)": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), None, dict(N=30, M=8, slack=True, n_decomp=12)),
    "cfg2 (N 20) for reference": (dict(N=20, S=5, n_lin=8, M=8), None, dict(N=20, M=8, tmpc_pp=True)),
}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, (dkw, okw, skw) in SHAPES.items():
    for nb in (5, 64):
        sc = scenes.make_scene(11, B=nb - 1 if skw.get("tmpc_pp") else nb, **skw)
        hx, h0, hp = sc["xinit"][:nb], sc["x0"][:nb], sc["params"][:nb]
        n = hx.shape[0]
        pb = O.problem(**(okw or dkw))
        xt, ut, o = O.solve_batch(pb, hx, h0.reshape(n, -1), hp.reshape(n, -1))
        s = solver.BatchedSolver(solver.default_dims(**dkw), B_max=n)
        row = {"shape": name, "planners": n, "by_mode": {}}
        for mode in (0, 1, 2, 3):
            if mode and not s.set_latency_mode(mode):
                continue
            s.set_latency_mode(mode)
            ts = []
            for _ in range(reps + 10):
                t1 = time.perf_counter()
                s.set_batch(hx, h0, hp); s.solve(sync=False); s.select_best()
                ts.append(time.perf_counter() - t1)
            s.get_timings(); s.enable_timing(32)
            for _ in range(20):
                s.solve(sync=False)
            k_ms = s.get_timings(); g = s.get()
            ok = o["exit_code"] == 1
            sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
            row["by_mode"][f"mode_{mode}"] = {
                "p50_ms": round(float(np.percentile(np.array(ts[10:]) * 1e3, 50)), 4), "kernel_ms": round(float(np.median(k_ms)), 4),
                "exit_code_mismatch": int((g["exit_code"] != o["exit_code"]).sum()), "sqp_iter_mismatch": int((g["sqp_iter"] != o["sqp_iter"]).sum()),
                "ipm_iter_mismatch": int((g["qp_iter_total"][ok] != o["qp_iter_total"][ok]).sum()),
                "max_rel": float((np.abs(g["xtraj"][ok] - xt[ok]) / sx).max()) if ok.any() else None}
        s.close()
        print(json.dumps(row), flush=True)
