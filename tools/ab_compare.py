"""A/B of two builds of the library (same C-ABI): bitwise comparison of every output on the BASELINE shapes + kernel time.

    python tools/ab_compare.py build/libtmpc_hip_prev.so [mpc_planner_amd/libtmpc_hip.so]

Used for scheduling-only kernel changes (operand prefetch, DPP variants): the results must be bit-identical."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from mpc_planner_amd import scenes, solver

lib_a = os.path.abspath(sys.argv[1])
lib_b = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else solver.LIB_PATH
SHAPES = (
    ("cfg1", dict(N=20, M=4, B=64, guidance=False), dict(N=20, S=5, n_lin=0, M=4), 64),
    ("cfg2", dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), 64),
    ("cfg4", dict(N=20, M=12, B=63, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12), 32),
    ("N30 two-wave", dict(N=30, M=8, B=64), dict(N=30, S=5, n_lin=8, M=8), 16),
    ("cfg3 two-wave", dict(N=30, M=8, B=64, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), 8),
    ("N21 odd horizon", dict(N=21, M=8, B=64), dict(N=21, S=5, n_lin=8, M=8), 4),
    ("cfg5", dict(N=20, M=8, B=32, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), 32),
)
for name, kw, dims_kw, n_scenes in SHAPES:
    batch = scenes.make_batch(range(700, 700 + n_scenes), **kw)
    B = batch["xinit"].shape[0]
    res, ms = [], []
    for lib in (lib_a, lib_b):
        s = solver.BatchedSolver(solver.default_dims(lib_path=lib, **dims_kw), B_max=B, lib_path=lib)
        s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); s.solve()
        ms.append(float(np.median(s.time_solve(5)))); res.append(s.get()); s.close()
    same = {k: bool(np.array_equal(res[0][k], res[1][k], equal_nan=True)) for k in res[0]}
    worst = float(np.nanmax(np.abs(res[0]["xtraj"] - res[1]["xtraj"]))) if not same["xtraj"] else 0.0
    print(json.dumps(dict(shape=name, B=B, bitwise_identical=all(same.values()), fields=same, max_abs_xtraj_diff=worst,
                          kernel_ms_a=ms[0], kernel_ms_b=ms[1], speedup=ms[0] / ms[1])), flush=True)
