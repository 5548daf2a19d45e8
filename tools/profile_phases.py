"""Print the in-kernel phase profile (mean shader cycles per trajectory) for a cfg2 batch.
Usage: python tools/profile_phases.py [B] [latency_mode]      latency_mode 0 (fast kernel of the shape), 1 (two waves), 2 (parallel in time)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_planner_amd import scenes, solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsc = max(B // 64, 1)
batch = scenes.make_batch(range(100, 100 + nsc), N=20, M=8, B=min(B, 64))
sv = solver.BatchedSolver(solver.default_dims(), B_max=batch["xinit"].shape[0])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sv.set_latency_mode(mode)
sv.set_batch(batch["xinit"], batch["x0"], batch["params"])
sv.solve()
prof = sv.debug_profile()
res = sv.get()
tot = prof["total"]
print(json.dumps({"B": int(batch["xinit"].shape[0]), "latency_mode": mode, "cycles": prof,
                  "frac": {k: round(v / tot, 4) for k, v in prof.items()},
                  "mean_sqp": float(res["sqp_iter"].mean()), "mean_ipm_total": float(res["qp_iter_total"].mean())}))
