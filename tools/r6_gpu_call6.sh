#!/bin/bash
# round 6, GPU call 6: the tick numbers of latency mode 3 beside modes 1 / 2 (cfg 2 tick of 64, the deployed 4 + 1, cfg 4's share of 8, cfg 5)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 5 --warmup 2 --no-end-to-end --no-tight --index-check-sets 16 --cpu-scenes 4 --latency-reps 200 > gpurun_out/r6_tick_bench.json 2> gpurun_out/r6_tick_bench.err
for m in 0 2 3; do
  timeout 600 python bench.py --workload cfg4 --share-of 8 --latency-mode $m --steps 50 --warmup 5 --no-cpu-baseline --latency-reps 0 > gpurun_out/r6_cfg4_share8_mode$m.json 2> gpurun_out/r6_cfg4_share8_mode$m.err
  timeout 600 python bench.py --workload cfg5 --latency-mode $m --steps 50 --warmup 5 --no-cpu-baseline --latency-reps 0 > gpurun_out/r6_cfg5_mode$m.json 2> gpurun_out/r6_cfg5_mode$m.err
done
python - <<'PY'
import json
def last(f):
    try: return json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: return None
d = last("gpurun_out/r6_tick_bench.json")
for key in ("latency_b64", "latency_b5"):
    b = d[key]
    print(key, "fastest", b["fastest_mode"], "p50", round(b["p50_ms"], 4), "kernel", round(b["kernel_ms"], 4))
    for m, v in b["by_mode"].items():
        print("   ", m, "p50", round(v["p50_ms"], 4), "p90", round(v["p90_ms"], 4), "kernel", round(v["kernel_ms"], 4), "parity", {k: v["parity"][k] for k in ("exit_code_mismatch", "sqp_iter_mismatch", "ipm_iter_mismatch", "parity_max_rel")})
for wl in ("cfg4_share8", "cfg5"):
    for m in (0, 2, 3):
        d = last(f"gpurun_out/r6_{wl}_mode{m}.json")
        if d: print(wl, "mode", m, "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 4), "value", round(d["value"]), "accepted", d["config"]["kernel_variant"]["accepted"], "parity", {k: d["parity"][k] for k in ("exit_code_mismatch", "sqp_iter_mismatch", "ipm_iter_mismatch", "parity_max_rel")} if d.get("parity") else None)
        else: print(wl, "mode", m, "FAILED")
PY
