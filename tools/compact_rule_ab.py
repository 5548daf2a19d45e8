"""One-wave shapes (N <= 21): the launch-size rule between the fast kernel (everything in LDS, four per CU) and the compact kernel (eight per
CU, persistent): bitwise comparison and kernel time per launch size.  A: TMPC_COMPACT_MIN_B=0 (compact for every launch, rounds 3-4);
B: the library's rule (fast kernel up to what it holds resident).
    python tools/compact_rule_ab.py > gpurun_out/compact_rule_ab.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_planner_amd import scenes

SHAPES = (
    ("cfg2 (8,8)", dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), 32, (64, 256, 512, 1024, 2048)),
    ("cfg4 (12,12)", dict(N=20, M=12, B=64), dict(N=20, S=5, n_lin=12, M=12), 32, (512, 1024, 2048)),
    ("cfg5 (24,0)", dict(N=20, M=8, B=32, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), 16, (32, 512)),
    ("cfg1 (0,4)", dict(N=20, M=4, B=64, guidance=False), dict(N=20, S=5, n_lin=0, M=4), 16, (64, 1024)),
)
BATCHES = {name: scenes.make_batch(range(900, 900 + n), workers=16, **kw) for name, kw, _, n, _ in SHAPES}    # (before the GPU runtime: forked workers)
print("scenes ready", file=sys.stderr, flush=True)
import torch
torch.cuda.init()
from mpc_planner_amd import solver
if not os.environ.get('TMPC_HIP_LIBRARY'):
    solver.LIB_PATH = solver.LAB_LIB_PATH      # (round 6: the TMPC_* kernel-selection switches exist in the lab build of the library only)
for name, kw, dims_kw, _, sizes in SHAPES:
    batch = BATCHES[name]
    for B in sizes:
        res, ms, info = [], [], []
        for env in (dict(TMPC_COMPACT_MIN_B="0"), {}):
            os.environ.pop("TMPC_COMPACT_MIN_B", None)
            os.environ.update(env)
            s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=B)
            s.set_batch(batch["xinit"][:B], batch["x0"][:B], batch["params"][:B]); s.solve(); s.solve()
            ms.append(float(np.median(s.time_solve(7)))); res.append(s.get()); info.append(s.kernel_info()); s.close()
        os.environ.pop("TMPC_COMPACT_MIN_B", None)
        same = {k: bool(np.array_equal(res[0][k], res[1][k], equal_nan=True)) for k in res[0]}
        print(json.dumps(dict(shape=name, B=B, bitwise_identical=all(same.values()), always_compact_ms=ms[0], rule_ms=ms[1], speedup=ms[0] / ms[1],
                              kernel_info=info[1] if B == sizes[0] else None)), flush=True)
