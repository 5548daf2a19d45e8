"""Latency variants on a SATURATED launch (they are built for a tick; this is the data point for DESIGN 8): solves/s of modes 0 / 1 / 2 at B = 8192."""
import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
from mpc_planner_amd import solver, scenes
dims = solver.default_dims(N=20)
parts = [scenes.make_scene(i, N=20, M=8, B=64) for i in range(8)]
rep = 16
xi = np.concatenate([p["xinit"] for p in parts] * rep); x0 = np.concatenate([p["x0"] for p in parts] * rep); pr = np.concatenate([p["params"] for p in parts] * rep)
B = xi.shape[0]
s = solver.BatchedSolver(dims, B_max=B)
for mode in (0, 1, 2):
    s.set_latency_mode(mode)
    s.set_batch(xi, x0, pr); s.solve(); s.solve()
    s.enable_timing(8)
    for _ in range(4): s.solve(sync=False)
    ms = float(np.median(s.get_timings()))
    r = s.get()
    print(json.dumps({"mode": mode, "B": B, "kernel_ms": ms, "solves_per_s": B / (ms * 1e-3), "success": float((r["exit_code"] == 1).mean())}), flush=True)
s.close()
