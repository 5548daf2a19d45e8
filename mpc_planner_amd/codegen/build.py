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
CSRC = os.path.join(os.path.dirname(HERE), "csrc", "tmpc_capi.hip")            # a generated solver is ONE translation unit: the C-ABI unit instantiates what it dispatches
CSRC_LANES = os.path.join(os.path.dirname(HERE), "csrc", "tmpc_lanes.hip")      # optional second unit (lane-per-trajectory kernels; TMPC_BUILD_LANES=1)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(header, out, extra):
    lanes = os.environ.get("TMPC_BUILD_LANES", "0") == "1"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-disable-machine-licm",
           "-Rpass-analysis=kernel-resource-usage", f'-DTMPC_GENERATED_STAGE="{header}"', *extra, *(["-DTMPC_WITH_LANES"] if lanes else []),
           "-o", out, CSRC, *([CSRC_LANES] if lanes else [])]
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


def wave_kernel_scratch(usage):
    """Scratch (bytes per lane) of the production wave kernels (tmpc_solve_kernel / tmpc_solve_fast_kernel<..., PROF = false>) of a build."""
    return {k: v.get("scratch", 0) for k, v in usage.items()
            if "tmpc_solve" in k and "lanes" not in k and "Lb1E" not in k and v.get("scratch", 0) > 0}


class GeneratedKernelSpills(RuntimeError):
    pass


def build_generated_solver(name, modules, model, settings, out_dir, method="symbolic", strict=False):
    """method: see emit.generate ("symbolic": best kernels, slow generation; "jets": instant generation).

    Zero-scratch rule for generated solvers (round-2 verdict): register spills inside partially masked regions have produced WRONG
    iterates in this kernel family, so a build whose production wave kernels use scratch is not taken silently: the other
    code-generation back end is tried (its emitted code has a different register profile), the build with zero scratch -- else the
    one with the least -- is kept, and the outcome is written to <name>_meta.json: "wave_kernel_scratch" (empty = clean) and
    "spill_free" (false = the library passed its parity tests, but nothing guarantees that for other inputs).
    strict=True raises GeneratedKernelSpills instead of keeping a spilling build."""
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(os.path.abspath(out_dir), f"libtmpc_hip_{name}.so")
    header = os.path.join(os.path.abspath(out_dir), f"stage_{name}.h")
    tried = {}
    best = None
    for m in (method, "jets" if method == "symbolic" else "symbolic"):
        gen = emit.generate(modules, model, settings, name, method=m)
        with open(header, "w") as fh:
            fh.write(gen["header"])
        # fast (register-row) kernels for the stack's row count: one-wave (N <= 21) and two-wave (22 <= N <= 32) variants
        tmp = out + f".{m}"
        usage = _compile(header, tmp, ["-DTMPC_GEN_FAST", "-DTMPC_GEN_FAST2"])
        spills = wave_kernel_scratch(usage)
        tried[m] = spills
        worst = max(spills.values(), default=0)
        if best is None or worst < best[0]:
            best = (worst, m, gen, usage, tmp)
        if not spills:
            break
    worst, m_used, gen, usage, tmp = best
    for m in tried:                                             # keep the chosen build (and the header that belongs to it)
        if m != m_used and os.path.exists(out + f".{m}"):
            os.remove(out + f".{m}")
    os.replace(tmp, out)
    with open(header, "w") as fh:
        fh.write(gen["header"])
    if worst > 0 and strict:
        raise GeneratedKernelSpills(f"{name}: production wave kernels use scratch with every back end: {tried}")
    meta = dict(name=name, npar=gen["npar"], nh=gen["nh"], slack=gen["slack"], rows=gen["rows"],
                parameter_map=dict(gen["params"]._params), kernel_resources=usage, codegen_method=m_used, codegen_methods_tried=tried,
                wave_kernel_scratch=wave_kernel_scratch(usage), spill_free=(worst == 0),
                note="hand-written shapes of libtmpc_hip.so are held to zero scratch.  Generated stage functions are straight-line "
                     "code from symbolic differentiation / jets and may spill inside the linearisation phase: a build whose wave "
                     "kernels use scratch is kept only after the other back end was tried, and is marked spill_free = "
                     "false (its parity tests pass, but spills under partial EXEC have produced wrong iterates in this kernel "
                     "family before: DESIGN 5)")
    with open(os.path.join(out_dir, f"{name}_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    return out, meta
