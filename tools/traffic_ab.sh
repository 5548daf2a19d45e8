#!/bin/bash
# Round-4 verdict next-5: does making the read-once / write-once global streams non-temporal keep the per-workgroup NLP workspace in the L2?
# Step 1 (here, no GPU): build the variants  build/exp/libtmpc_hip_nt{1,2}.so  (same sources, -DTMPC_EXP_NT=1 / 2: csrc/tmpc_stage.hpp).
# Step 2 (GPU box):  tools/traffic_ab.sh measure  -> gpurun_out/round5_c_traffic_nontemporal_ab.json
#   per variant: kernel time (bench.py, HIP events) + rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs), calibrated bytes per count.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
if [ "$1" != "measure" ]; then
  mkdir -p build/exp
  for v in 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -disable-machine-licm -DTMPC_TU_COMPACT -DTMPC_EXP_NT=$v \
        -o build/exp/tmpc_solve_nt$v.o mpc_planner_amd/csrc/tmpc_solve.hip &      # (the compact one-wave kernels: what the cfg 2 bench launch runs)
  done
  wait
  for v in 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libtmpc_hip_nt$v.so build/exp/tmpc_solve_nt$v.o build/obj/tmpc_solve_fast.o build/obj/tmpc_solve_prof.o build/obj/tmpc_solve_cp2.o build/obj/tmpc_capi.o
  done
  ls -la build/exp/*.so
  exit 0
fi
O=$R/gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import json, os, subprocess, sys, glob, csv, shutil
R = os.getcwd(); O = os.path.join(R, "gpurun_out")
BENCH = [sys.executable, os.path.join(R, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--latency-reps", "0", "--no-tight", "--no-end-to-end",
         "--parity-check", "64", "--index-check-sets", "0", "--gen-workers", "1", "--scene-cache", "/tmp/tmpc_bench_scenes"]
out = {"what": "cfg 2 bench launch (32768 trajectories), compact kernel: plain vs non-temporal accesses to the read-once / write-once global streams "
               "(TMPC_EXP_NT, csrc/tmpc_stage.hpp); FETCH_SIZE / WRITE_SIZE per launch from separate rocprofv3 --pmc passes", "variants": {}}
subprocess.run(BENCH + ["--steps", "1", "--warmup", "0"], cwd="/tmp", capture_output=True, text=True, timeout=600)     # fills the scene cache
# calibration of the counters' units (known 1 GiB streams), as tools/collect_profiles.py does
cal = {}
calib = os.path.join(R, "build", "tools", "pmc_calibrate")
if not os.path.exists(calib):
    os.makedirs(os.path.dirname(calib), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", calib, os.path.join(R, "tools", "pmc_calibrate.hip")], check=True)
def pmc(counter, cmd, env, tag):
    d = os.path.join(O, f"r5c_{tag}_{counter}")
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--pmc", counter, "-d", d, "--output-format", "csv", "--"] + cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter]
    shutil.rmtree(d, ignore_errors=True)
    return rows
env0 = dict(os.environ, TMPDIR="/tmp")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = pmc(c, [calib], env0, "cal")
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    cal[c] = {k: sum(v) / len(v) for k, v in by.items()}
out["calibration_counts"] = cal
GIB = float(1 << 30)
def unit(counter, kernel_sub):
    for k, v in cal.get(counter, {}).items():
        if kernel_sub in k and v > 0:
            return GIB / v
    return None
u_fetch = unit("FETCH_SIZE", "read")
u_write = unit("WRITE_SIZE", "write")
out["bytes_per_count"] = {"FETCH_SIZE": u_fetch, "WRITE_SIZE": u_write}
for name, lib in (("plain", None), ("nt_io", "build/exp/libtmpc_hip_nt1.so"), ("nt_io_and_params", "build/exp/libtmpc_hip_nt2.so")):
    env = dict(env0)
    if lib:
        env["TMPC_HIP_LIBRARY"] = os.path.join(R, lib)
    r = subprocess.run(BENCH, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rec = {}
    if line:
        d = json.loads(line[-1])
        rec = {"kernel_ms_avg": d["roofline"]["kernel_ms_avg"], "value": d["value"], "parity": {k: d["parity"][k] for k in ("exit_code_mismatch", "ipm_iter_mismatch", "parity_max_rel")}}
    else:
        rec = {"error": r.stderr[-500:]}
    for c, u in (("FETCH_SIZE", u_fetch), ("WRITE_SIZE", u_write)):
        rows = [x for x in pmc(c, BENCH, env, name) if "tmpc_solve_compact" in x["Kernel_Name"]]
        vals = [float(x["Counter_Value"]) for x in rows]
        if vals:
            per = sum(vals) / len(vals)
            rec[c] = {"counts_per_launch": per, "bytes_per_launch": per * u if u else None, "KB_per_solve": per * u / 32768 / 1e3 if u else None, "launches": len(vals)}
    out["variants"][name] = rec
    print(name, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(O, "round5_c_traffic_nontemporal_ab.json"), "w"), indent=1)
PY
