"""GPU: throughput (lane-per-trajectory) variant vs the default wave-per-trajectory kernels on cfg 2 -- parity of the two against
each other on the same batch, and kernel time per launch at several batch sizes.
Usage: python tools/bench_lanes.py [scenes ...]        (scenes per launch, default 256 1024)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mpc_planner_amd import scenes, solver  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 1024]
uniq = 128
t0 = time.time()
base = scenes.make_batch(range(0, uniq), N=20, M=8, B=64)
print(f"# {uniq} scenes generated in {time.time() - t0:.1f} s", flush=True)
LIB = os.environ.get("TMPC_LIB")          # experiments: an alternative build of the library
dims = solver.default_dims(N=20, S=5, n_lin=8, M=8, lib_path=LIB)
for n_sc in sizes:
    reps = (n_sc + uniq - 1) // uniq
    xi = np.concatenate([base["xinit"]] * reps)[:n_sc * 64]; x0 = np.concatenate([base["x0"]] * reps)[:n_sc * 64]
    pa = np.concatenate([base["params"]] * reps)[:n_sc * 64]
    B = xi.shape[0]
    dev = torch.device("cuda")
    t_xi = torch.from_numpy(xi).to(dev); t_x0 = torch.from_numpy(x0.reshape(B, -1)).to(dev); t_pa = torch.from_numpy(pa.reshape(B, -1)).to(dev)
    out = {"scenes": n_sc, "B": B}
    res = {}
    for mode in ("wave", "lanes"):
        s = solver.BatchedSolver(dims, B_max=B, lib_path=LIB)
        if mode == "lanes":
            s.set_throughput_mode(True)
        s.set_batch_device(B, t_xi.data_ptr(), t_x0.data_ptr(), t_pa.data_ptr())
        s.solve(); s.solve()
        ms = s.time_solve(5)
        res[mode] = s.get()
        out[mode + "_ms"] = float(np.median(ms)); out[mode + "_solves_per_s"] = float(B / (np.median(ms) * 1e-3))
        s.close()
    a, b = res["wave"], res["lanes"]
    ok = (a["exit_code"] == 1) & (b["exit_code"] == 1)
    out.update(exit_mismatch=int((a["exit_code"] != b["exit_code"]).sum()), sqp_mismatch=int((a["sqp_iter"] != b["sqp_iter"]).sum()),
               ipm_mismatch=int((a["qp_iter_total"][ok] != b["qp_iter_total"][ok]).sum()),
               max_abs_traj_diff=float(np.abs(a["xtraj"][ok] - b["xtraj"][ok]).max()), success=float((b["exit_code"] == 1).mean()),
               mean_ipm_per_solve=float(b["qp_iter_total"].mean()))
    print(json.dumps(out), flush=True)
