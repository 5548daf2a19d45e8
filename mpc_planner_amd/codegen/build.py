"""Build a per-configuration solver library from a module stack: emit the stage functions, compile the solve kernels
around them (same sources, -DTMPC_GENERATED_STAGE), report what the compiler made of it.

    path, meta = build_generated_solver("my_cfg", modules, model, settings, out_dir)
    dims = solver.default_dims(N=20, lib_path=path); s = solver.BatchedSolver(dims, B_max=64, lib_path=path)

The library exports the C-ABI of include/tmpc_hip.h; its parameter row layout is the module stack's
(meta["parameter_map"], the same order the reference's generator would produce for the same modules).
"""
import json
import os
import re
import subprocess

from . import emit

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc", "tmpc_solve.hip")
CSRC_LANES = os.path.join(os.path.dirname(HERE), "csrc", "tmpc_lanes.hip")      # second translation unit (throughput kernels)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(header, out, extra):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-disable-machine-licm",
           "-Rpass-analysis=kernel-resource-usage", f'-DTMPC_GENERATED_STAGE="{header}"', *extra, "-o", out, CSRC, CSRC_LANES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    usage, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and name and ("tmpc_solve" in name or "lanes_solve" in name):
                usage.setdefault(name, {})[key] = int(m.group(1))
    return usage


def build_generated_solver(name, modules, model, settings, out_dir, method="symbolic"):
    """method: see emit.generate ("symbolic": best kernels, slow generation; "jets": instant generation)."""
    os.makedirs(out_dir, exist_ok=True)
    gen = emit.generate(modules, model, settings, name, method=method)
    header = os.path.join(os.path.abspath(out_dir), f"stage_{name}.h")
    with open(header, "w") as fh:
        fh.write(gen["header"])
    out = os.path.join(os.path.abspath(out_dir), f"libtmpc_hip_{name}.so")
    # fast (register-row) kernels for the stack's row count: one-wave (N <= 21) and two-wave (22 <= N <= 32) variants
    usage = _compile(header, out, ["-DTMPC_GEN_FAST", "-DTMPC_GEN_FAST2"])
    meta = dict(name=name, npar=gen["npar"], nh=gen["nh"], slack=gen["slack"], rows=gen["rows"],
                parameter_map=dict(gen["params"]._params), kernel_resources=usage,
                note="hand-written shapes of libtmpc_hip.so are held to zero scratch; generated stage functions are "
                     "straight-line code from symbolic differentiation and may spill inside the linearisation phase "
                     "(see kernel_resources[*].scratch); results are checked against the hand-written kernels in tests")
    with open(os.path.join(out_dir, f"{name}_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    return out, meta
