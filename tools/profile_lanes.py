"""GPU: phase profile of the lane-per-trajectory kernel.  Builds a profiling twin of the library (-DTMPC_LANES_PROF: shader-clock
reads around linearise and the four interior-point sweeps, accumulated per lane), runs cfg 2 and prints mean cycles per phase.
Usage (on the GPU box): python tools/profile_lanes.py [scenes]        -> JSON line"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "libtmpc_hip_prof.so")


def build():
    csrc = os.path.join(ROOT, "mpc_planner_amd", "csrc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = os.path.join(ROOT, "build", "obj", "tmpc_lanes_prof.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-mllvm", "-disable-machine-licm",
                           "-DTMPC_LANES_PROF", "-o", obj, os.path.join(csrc, "tmpc_lanes.hip")])
    o = os.path.join(ROOT, "build", "obj")                                       # (needs a build with the lane kernels: TMPC_BUILD_LANES=1)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT,
                           *[os.path.join(o, f"{u}.o") for u in ("tmpc_solve_fast", "tmpc_solve_compact", "tmpc_solve_prof", "tmpc_solve_cp2", "tmpc_capi_lanes")], obj])


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    import torch
    from mpc_planner_amd import scenes, solver
    n_sc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    uniq = min(n_sc, 64)
    base = scenes.make_batch(range(0, uniq), N=20, M=8, B=64)
    reps = (n_sc + uniq - 1) // uniq
    xi = np.concatenate([base["xinit"]] * reps)[:n_sc * 64]; x0 = np.concatenate([base["x0"]] * reps)[:n_sc * 64]
    pa = np.concatenate([base["params"]] * reps)[:n_sc * 64]
    B = xi.shape[0]
    dims = solver.default_dims(N=20, S=5, n_lin=8, M=8, lib_path=OUT)
    s = solver.BatchedSolver(dims, B_max=B, lib_path=OUT)
    s.set_throughput_mode(True)
    s.set_batch(xi, x0, pa)
    s.solve(); ms = s.time_solve(3)
    cyc = np.zeros(7)
    lib = solver.load_library(OUT)
    lib.tmpc_lanes_debug_profile.argtypes = [C.c_int, C.c_void_p]
    assert lib.tmpc_lanes_debug_profile(min(B, 65536), cyc.ctypes.data_as(C.c_void_p)) == 0
    g = s.get()
    names = ["linearise", "sweep_A", "sweep_B", "sweep_C", "sweep_D", "step", "final"]
    tot = cyc.sum()
    print(json.dumps({"B": B, "kernel_ms": float(np.median(ms)), "mean_cycles_per_solve": {n: float(c) for n, c in zip(names, cyc)},
                      "fraction": {n: float(c / tot) for n, c in zip(names, cyc)}, "mean_ipm_iter": float(g["qp_iter_total"].mean()),
                      "cycles_per_stage_visit": {n: float(c / (g["qp_iter_total"].mean() + (10 if n == "sweep_A" else 0)) / 21) for n, c in zip(names[1:5], cyc[1:5])}}))
