"""Run on the GPU box (gpurun): collect the rocprofv3 evidence bench.py's roofline refers to and write one summary JSON.
  pass 1  rocprofv3 --kernel-trace --stats         (per-kernel durations of the bench command)
  pass 2+ rocprofv3 --pmc <set>                    (one counter set per pass; never combined with tracing)
  calib   tools/pmc_calibrate.hip under --pmc FETCH_SIZE / WRITE_SIZE (known 1 GiB streams, 8 B per lane)
Usage: python tools/collect_profiles.py <tag> [extra bench.py arguments]      -> gpurun_out/<tag>_rocprof_summary.json, <tag>_kernel_stats.csv"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "profile"
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--latency-reps", "0", "--no-lanes", "--no-tight", "--parity-check", "0", "--gen-workers", "1", "--scene-cache", "/tmp/tmpc_bench_scenes"] + sys.argv[2:]
env = dict(os.environ, TMPDIR="/tmp")


def rocprof(name, args, cmd):
    d = os.path.join(OUT, f"{tag}_{name}")
    shutil.rmtree(d, ignore_errors=True)
    try:
        r = subprocess.run(["rocprofv3", *args, "-d", d, "--output-format", "csv", "--"] + cmd, cwd="/tmp", env=env,
                           capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            print(name, "failed:", r.stderr[-2000:], flush=True)
    except subprocess.TimeoutExpired:
        print(name, "timed out after 240 s", flush=True)
    return d


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", f"*{suffix}.csv"), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


sys.path.insert(0, ROOT)
import bench as _bench                                   # library_sha256(): hash of the kernel sources, what bench.py matches on
LIB_HASH = _bench.library_sha256()
subprocess.run(BENCH + ["--steps", "1", "--warmup", "0"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)   # fills the scene cache
summary = {"command": "rocprofv3 --kernel-trace --stats -- " + " ".join(BENCH[1:]) + "   (cfg2, bench.py's default launch: 512 scenes x 64 = 32768 trajectories); "
           "PMC in separate --pmc passes; kernel trace and every counter pass ran on the same build of the library", "tag": tag,
           "library_sha256": LIB_HASH, "trajectories_per_launch": 512 * 64}
d = rocprof("kt", ["--kernel-trace", "--stats"], BENCH)
ks = rows(d, "kernel_stats")
summary["kernel_stats"] = ks
if ks:
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(ks[0].keys())); w.writeheader(); w.writerows(ks)

SETS = [["FETCH_SIZE"], ["WRITE_SIZE"],
        ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
        ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"]]
def flush():
    with open(os.path.join(OUT, f"{tag}_rocprof_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


flush()
pmc = {}
for i, cs in enumerate(SETS):
    d = rocprof(f"pmc{i}", ["--pmc", *cs], BENCH)
    acc = {}
    for r in rows(d, "counter_collection"):
        if "tmpc_solve" not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
    summary["pmc_mean_per_launch"] = pmc
    flush()

# calibration of FETCH_SIZE / WRITE_SIZE on a known stream with the solve kernel's access width
exe = "/tmp/pmc_calibrate"
cal = {}
if subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "pmc_calibrate.hip")]).returncode == 0:
    for cname, kern in (("FETCH_SIZE", "calib_read"), ("WRITE_SIZE", "calib_write")):
        d = rocprof(f"cal_{cname}", ["--pmc", cname], [exe])
        vals = [float(r["Counter_Value"]) for r in rows(d, "counter_collection") if kern in r.get("Kernel_Name", "") and r["Counter_Name"] == cname]
        if vals:
            cal[cname] = dict(kernel=kern, counter_mean=sum(vals) / len(vals), true_bytes=float(1 << 30))
            cal[cname]["bytes_per_count"] = cal[cname]["true_bytes"] / cal[cname]["counter_mean"]
summary["calibration"] = cal
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fb = cal.get("FETCH_SIZE", {}).get("bytes_per_count", 1024.0)
    wb = cal.get("WRITE_SIZE", {}).get("bytes_per_count", 1024.0)
    summary["hbm_bytes_per_launch"] = dict(fetch=pmc["FETCH_SIZE"] * fb, write=pmc["WRITE_SIZE"] * wb,
                                           total=pmc["FETCH_SIZE"] * fb + pmc["WRITE_SIZE"] * wb,
                                           note="counter x calibrated bytes-per-count (known 1 GiB read / write streams, 8 B per lane)")
with open(os.path.join(OUT, f"{tag}_rocprof_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
if "hbm_bytes_per_launch" in summary:                        # what bench.py's roofline.traffic reads (matched by library hash + launch size)
    with open(os.path.join(OUT, f"{tag}_pmc.json"), "w") as fh:
        json.dump({"library_sha256": LIB_HASH, "trajectories_per_launch": 512 * 64, "hbm_bytes_per_launch": summary["hbm_bytes_per_launch"]["total"],
                   "fetch": summary["hbm_bytes_per_launch"]["fetch"], "write": summary["hbm_bytes_per_launch"]["write"],
                   "extra_bench_args": sys.argv[2:],                        # e.g. ["--no-param-sharing"]: bench.py matches the configuration too
                   "source": f"tools/collect_profiles.py {tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the solve kernel, calibrated bytes per count"}, fh, indent=1)
print(json.dumps({k: summary[k] for k in ("calibration", "pmc_mean_per_launch") if k in summary}, indent=1))
for r in ks[:4]:
    print(r.get("Name", "")[:80], r.get("Calls"), r.get("AverageNs"))
for sub in glob.glob(os.path.join(OUT, f"{tag}_*")):
    if os.path.isdir(sub):
        shutil.rmtree(sub, ignore_errors=True)          # raw traces are large; the summary is what gets committed
