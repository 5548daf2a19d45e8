#!/bin/bash
# round 6, GPU call 3: the four-wave tick kernel (latency mode 3): parity, phases, tick latency vs modes 1 / 2
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_quad.py -x -q -s > gpurun_out/r6_quad_tests.log 2>&1
tail -5 gpurun_out/r6_quad_tests.log
for m in 1 2 3; do python tools/profile_phases.py 64 $m; done > gpurun_out/r6_quad_phases.jsonl 2> gpurun_out/r6_quad_phases.err
cat gpurun_out/r6_quad_phases.jsonl | cut -c1-900
timeout 1200 python bench.py --steps 5 --warmup 2 --no-end-to-end --no-tight --index-check-sets 16 --cpu-scenes 4 > gpurun_out/r6_quad_bench.json 2> gpurun_out/r6_quad_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_quad_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
for key in ("latency_b64", "latency_b5"):
    b = d[key]
    print(key, "fastest", b["fastest_mode"], "p50", b["p50_ms"], "kernel", b["kernel_ms"])
    for m, v in b["by_mode"].items():
        print("   ", m, "p50", round(v["p50_ms"], 4), "p90", round(v["p90_ms"], 4), "kernel", round(v["kernel_ms"], 4), "parity", {k: v["parity"][k] for k in ("exit_code_mismatch", "sqp_iter_mismatch", "ipm_iter_mismatch", "parity_max_rel")})
    print("    vs_mode_1", b.get("vs_mode_1"))
PY
