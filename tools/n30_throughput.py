"""Saturated throughput of the jackalsimulator T-MPC stack at the horizon it ships with (8 + 8 rows, N = 30; not a BASELINE config): 256 sets x 64 trajectories per launch.
    python tools/n30_throughput.py          (lab library + TMPC_NO_ONE_WAVE_N30=1: the two-wave kernels of rounds 3-5)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from mpc_planner_amd import scenes, solver
import oracle_lib as O
b = scenes.make_batch(range(300, 316), N=30, M=8, B=64)
rep = 16
big = [np.tile(b[k], (rep,) + (1,) * (b[k].ndim - 1)) for k in ("xinit", "x0", "params")]
B = big[0].shape[0]
s = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8), B_max=B)
s.set_batch(*big)
for _ in range(3):
    s.solve()
s.enable_timing(32)
for _ in range(10):
    s.solve(sync=False)
ms = float(np.median(s.get_timings())); g = s.get()
n = 128
pb = O.problem(N=30, S=5, n_lin=8, M=8)
xt, ut, o = O.solve_batch(pb, b["xinit"][:n], b["x0"][:n].reshape(n, -1), b["params"][:n].reshape(n, -1))
ok = o["exit_code"] == 1
print(json.dumps({"trajectories_per_launch": int(B), "kernel_ms": ms, "solves_per_s": B * float((g["exit_code"] == 1).mean()) / (ms * 1e-3), "kernel": s.kernel_info()[:160],
                  "parity_128": {"exit_code_mismatch": int((g["exit_code"][:n] != o["exit_code"]).sum()), "ipm_iter_mismatch": int((g["qp_iter_total"][:n][ok] != o["qp_iter_total"][ok]).sum()),
                                 "max_rel": float((np.abs(g["xtraj"][:n][ok] - xt[ok]) / np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)).max())}}))
s.close()
