"""PCIe-inclusive rate of the bench shape: host buffers in (tmpc_set_batch: xinit, warm start, parameters), solve, results
out (tmpc_get), 4096 trajectories per launch -- the number DESIGN.md quotes next to the HBM-resident bench value."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_planner_amd import scenes, solver
batch = scenes.make_batch(range(64), N=20, M=8, B=64)
B = batch["xinit"].shape[0]
s = solver.BatchedSolver(solver.default_dims(), B_max=B)
for _ in range(3):
    s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); s.get()
ts = {"h2d": [], "solve": [], "d2h": [], "total": []}
for _ in range(10):
    t0 = time.perf_counter(); s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.synchronize()
    t1 = time.perf_counter(); s.solve(); t2 = time.perf_counter(); s.get(); t3 = time.perf_counter()
    ts["h2d"].append(t1 - t0); ts["solve"].append(t2 - t1); ts["d2h"].append(t3 - t2); ts["total"].append(t3 - t0)
med = {k: float(np.median(v)) * 1e3 for k, v in ts.items()}
nbytes = batch["xinit"].nbytes + batch["x0"].nbytes + batch["params"].nbytes
print(json.dumps(dict(B=B, ms=med, h2d_GBps=nbytes / (med["h2d"] * 1e-3) / 1e9, host_input_MB=nbytes / 1e6,
                      solves_per_s_pcie_inclusive=B / (med["total"] * 1e-3), note="pageable numpy buffers, no overlap of copy and solve")))
