"""Coarse shader-clock split of the Riccati sweeps (factor: terminal block / stage loop; solve: stage-parallel prologue,
backward sweep, forward sweep, stage-parallel epilogue) for a cfg 2 batch.

Builds a SEPARATE library, build/libtmpc_hip_sweepprof.so, from the same sources with -DTMPC_SWEEP_PROFILE (the product
library never carries the clock reads); `--build-only` compiles it (no GPU needed), without the flag the library must exist.
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "build", "libtmpc_hip_sweepprof.so")
NAMES = ["factor_terminal", "factor_loop", "solve_prologue", "solve_backward", "solve_forward", "solve_epilogue",
         "calls_factor", "calls_solve"]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "mpc_planner_amd", "csrc", "tmpc_capi.hip")        # one translation unit (-DTMPC_SINGLE_TU: the C-ABI unit instantiates what it dispatches)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-shared", "-mllvm", "-disable-machine-licm", "-DTMPC_SWEEP_PROFILE", "-DTMPC_SINGLE_TU", "-o", LIB, src])


def main():
    if "--build-only" in sys.argv:
        return build()
    import torch
    torch.cuda.init()
    from mpc_planner_amd import scenes, solver
    out = {}
    for B in (64, 4096):
        batch = scenes.make_batch(range(100, 100 + B // 64), N=20, M=8, B=64)
        dims = solver.default_dims(lib_path=LIB)
        sv = solver.BatchedSolver(dims, B_max=B, lib_path=LIB)
        sv.lib.tmpc_debug_sweep_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        buf = (C.c_uint64 * 8)()
        sv.set_batch(batch["xinit"], batch["x0"], batch["params"])
        sv.solve()                                                     # warm-up
        sv.lib.tmpc_debug_sweep_profile(sv._h, buf, 8)                 # reset
        sv.solve()
        assert sv.lib.tmpc_debug_sweep_profile(sv._h, buf, 8) == 0
        v = dict(zip(NAMES, [int(x) for x in buf]))
        res = sv.get()
        nf, ns = max(v["calls_factor"], 1), max(v["calls_solve"], 1)
        out[f"B{B}"] = {"cycles_total": v,
                        "per_call": {"factor_terminal": v["factor_terminal"] / nf, "factor_loop": v["factor_loop"] / nf,
                                     "factor_loop_per_stage": v["factor_loop"] / nf / 20,
                                     "solve_prologue": v["solve_prologue"] / ns, "solve_backward": v["solve_backward"] / ns,
                                     "solve_backward_per_stage": v["solve_backward"] / ns / 20,
                                     "solve_forward": v["solve_forward"] / ns, "solve_forward_per_stage": v["solve_forward"] / ns / 20,
                                     "solve_epilogue": v["solve_epilogue"] / ns},
                        "mean_ipm_total": float(res["qp_iter_total"].mean()),
                        "note": "clock64() shader-clock ticks (s_memtime), summed by lane 0 of every trajectory"}
        sv.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
