"""A/B of the compact kernel (two waves per SIMD, eight trajectories per CU) against the fast kernel of the same shape and against a
reference build of the library: bitwise comparison of every output + kernel time, and the throughput as a function of the resident
workgroups per CU (TMPC_COMPACT_PER_CU).

    python tools/ab_compact.py [reference.so] [--scenes 512] > gpurun_out/ab_compact.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from mpc_planner_amd import scenes, solver
if not os.environ.get('TMPC_HIP_LIBRARY'):
    solver.LIB_PATH = solver.LAB_LIB_PATH      # (round 6: the TMPC_* kernel-selection switches exist in the lab build of the library only)

ap = argparse.ArgumentParser()
ap.add_argument("ref", nargs="?", default="")
ap.add_argument("--scenes", type=int, default=512)
ap.add_argument("--per-cu", default="8")
a = ap.parse_args()
dims_kw = dict(N=20, S=5, n_lin=8, M=8)


def run(lib, env, batch, reps=5):
    for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT_PER_CU"):
        os.environ.pop(k, None)
    os.environ.update(env)
    B = batch["xinit"].shape[0]
    s = solver.BatchedSolver(solver.default_dims(lib_path=lib, **dims_kw), B_max=B, lib_path=lib)
    s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); s.solve()
    ms = float(np.median(s.time_solve(reps))); res = s.get(); s.close()
    return res, ms


small = scenes.make_batch(range(700, 764), N=20, M=8, B=64)
variants = [("new_fast", solver.LIB_PATH, {"TMPC_NO_COMPACT": "1"}), ("new_compact", solver.LIB_PATH, {})]
if a.ref:
    variants.insert(0, ("ref", os.path.abspath(a.ref), {}))
base = None
for name, lib, env in variants:
    res, ms = run(lib, env, small)
    if base is None:
        base = res
    same = {k: bool(np.array_equal(base[k], res[k], equal_nan=True)) for k in res}
    print(json.dumps(dict(what="bitwise vs first", variant=name, B=int(small["xinit"].shape[0]), identical=all(same.values()), fields=same,
                          max_abs_xtraj_diff=float(np.nanmax(np.abs(base["xtraj"] - res["xtraj"]))), kernel_ms=ms,
                          success=float((res["exit_code"] == 1).mean()))), flush=True)
big = scenes.make_batch(range(0, a.scenes), N=20, M=8, B=64, workers=16)
B = big["xinit"].shape[0]
r0, ms0 = run(solver.LIB_PATH, {"TMPC_NO_COMPACT": "1"}, big, reps=3)
print(json.dumps(dict(what="throughput", variant="new_fast", B=B, kernel_ms=ms0, solves_per_s=B / ms0 * 1e3)), flush=True)
rc = None
for pc in a.per_cu.split(","):
    if True:
        r1, ms1 = run(solver.LIB_PATH, {"TMPC_COMPACT_PER_CU": pc}, big, reps=3)
        if rc is None:
            rc = r1
        same = all(np.array_equal(rc[k], r1[k], equal_nan=True) for k in rc)
        print(json.dumps(dict(what="throughput", variant=f"compact, workgroups per CU <= {pc}", B=B, kernel_ms=ms1, solves_per_s=B / ms1 * 1e3,
                              identical_to_first_compact=bool(same), exit_codes_equal_fast=bool(np.array_equal(r0["exit_code"], r1["exit_code"])),
                              ipm_iters_equal_fast=bool(np.array_equal(r0["qp_iter_total"], r1["qp_iter_total"])), speedup_vs_fast=ms0 / ms1)), flush=True)
