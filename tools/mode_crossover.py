"""Which kernel variant for which launch size (cfg 2): kernel time of modes 0 / 1 / 2 at B = 64 .. 4096 -> one JSON line per size."""
import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
from mpc_planner_amd import solver, scenes
dims = solver.default_dims(N=20)
parts = [scenes.make_scene(i, N=20, M=8, B=64) for i in range(16)]
xi = np.concatenate([p["xinit"] for p in parts] * 4); x0 = np.concatenate([p["x0"] for p in parts] * 4); pr = np.concatenate([p["params"] for p in parts] * 4)
s = solver.BatchedSolver(dims, B_max=4096)
for B in (64, 128, 256, 512, 1024, 2048, 4096):
    rec = {"B": B}
    for mode in (0, 1, 2):
        s.set_latency_mode(mode)
        s.set_batch(xi[:B], x0[:B], pr[:B]); s.solve(); s.solve()
        s.enable_timing(16)
        for _ in range(8): s.solve(sync=False)
        rec[f"mode{mode}_ms"] = round(float(np.median(s.get_timings())), 3)
    print(json.dumps(rec), flush=True)
s.close()
