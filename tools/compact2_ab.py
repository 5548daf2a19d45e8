"""Two-wave compact kernel (22 <= N <= 32, four trajectories per CU) against the fast two-wave kernel (two per CU) of the same shape:
bitwise comparison of every output and kernel time at several launch sizes.  A: TMPC_NO_COMPACT=1 (fast kernel whatever the launch size);
B: TMPC_COMPACT2_MIN_B=0 (compact kernel whatever the launch size).  The library's own rule (compact above what the fast kernel holds
resident) is printed from tmpc_kernel_info.
    python tools/compact2_ab.py [scenes_per_shape] > gpurun_out/compact2_ab.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_planner_amd import scenes

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = (
    ("cfg3 CA-MPC (20,8) CM=1", dict(N=30, M=8, B=64, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1)),
    ("cfg3 MPCC (20,8)", dict(N=30, M=8, B=64, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1)),
    ("N=30 (8,8)", dict(N=30, M=8, B=64), dict(N=30, S=5, n_lin=8, M=8)),
    ("N=30 (12,12)", dict(N=30, M=12, B=64), dict(N=30, S=5, n_lin=12, M=12)),
    ("jackal default: Gaussian rows, runtime shape (5+5) CM=2", dict(N=30, M=5, S=3, B=64, chance=True), dict(N=30, S=3, n_lin=5, M=5, row_model=1)),
    ("N=30 runtime shape (10+10)", dict(N=30, M=10, B=64), dict(N=30, S=5, n_lin=10, M=10)),
    ("N=22 (8,8)", dict(N=22, M=8, B=64), dict(N=22, S=5, n_lin=8, M=8)),
    ("N=32 (8,8)", dict(N=32, M=8, B=64), dict(N=32, S=5, n_lin=8, M=8)),
)
# every scene first (forked workers), the GPU runtime only afterwards
def is_small(name):
    return name.startswith("N=22") or name.startswith("N=32") or "runtime shape (10" in name
BATCHES = {name: scenes.make_batch(range(800, 800 + (8 if is_small(name) else n_scenes)), workers=16, **kw) for name, kw, _ in SHAPES}
print("scenes ready", file=sys.stderr, flush=True)
import torch
torch.cuda.init()
from mpc_planner_amd import solver
if not os.environ.get('TMPC_HIP_LIBRARY'):
    solver.LIB_PATH = solver.LAB_LIB_PATH      # (round 6: the TMPC_* kernel-selection switches exist in the lab build of the library only)
for name, kw, dims_kw in SHAPES:
    small = is_small(name)
    batch = BATCHES[name]
    Bfull = batch["xinit"].shape[0]
    for B in ((Bfull,) if small else (512, 1024, Bfull)):
        res, ms, info = [], [], []
        for env in (dict(TMPC_NO_COMPACT="1"), dict(TMPC_COMPACT2_MIN_B="0")):
            for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT2_MIN_B"):
                os.environ.pop(k, None)
            os.environ.update(env)
            s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=B)
            s.set_batch(batch["xinit"][:B], batch["x0"][:B], batch["params"][:B]); s.solve(); s.solve()
            ms.append(float(np.median(s.time_solve(5)))); res.append(s.get()); info.append(s.kernel_info()); s.close()
        same = {k: bool(np.array_equal(res[0][k], res[1][k], equal_nan=True)) for k in res[0]}
        print(json.dumps(dict(shape=name, B=B, bitwise_identical=all(same.values()), fields={k: v for k, v in same.items() if not v},
                              success=float((res[1]["exit_code"] == 1).mean()), fast_two_wave_ms=ms[0], compact_two_wave_ms=ms[1], speedup=ms[0] / ms[1],
                              solves_per_s_fast=B / ms[0] * 1e3, solves_per_s_compact=B / ms[1] * 1e3, kernel_info=info[1] if B == Bfull else None)), flush=True)
for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT2_MIN_B"):
    os.environ.pop(k, None)
