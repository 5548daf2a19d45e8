#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lds_poison.py tests/test_gpu_quad.py -x -q > gpurun_out/r6_poison_tests.log 2>&1; tail -15 gpurun_out/r6_poison_tests.log
for m in 2 3; do
  timeout 600 python bench.py --workload cfg4 --share-of 8 --latency-mode $m --steps 50 --warmup 5 --no-cpu-baseline --latency-reps 0 > gpurun_out/r6_cfg4_share8_mode$m.json 2> gpurun_out/r6_cfg4_share8_mode$m.err
  timeout 600 python bench.py --workload cfg5 --latency-mode $m --steps 50 --warmup 5 --no-cpu-baseline --latency-reps 0 > gpurun_out/r6_cfg5_mode$m.json 2> gpurun_out/r6_cfg5_mode$m.err
done
python - <<'PY'
import json
def last(f):
    try: return json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e: return None
for wl in ("cfg4_share8", "cfg5"):
    for m in (2, 3):
        d = last(f"gpurun_out/r6_{wl}_mode{m}.json")
        if d: print(wl, "mode", m, "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 4), "value", round(d["value"]), "success", d["config"]["success_fraction"], "parity", {k: d["parity"][k] for k in ("exit_code_mismatch", "sqp_iter_mismatch", "ipm_iter_mismatch", "parity_max_rel")} if d.get("parity") else None)
        else: print(wl, "mode", m, "FAILED")
PY
