#!/bin/bash
# two-wave compact kernels: tests, then the N = 30 shapes on a saturated GPU
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_compact2.py -x -q > $O/compact2_tests.log 2>&1 < /dev/null; tail -5 $O/compact2_tests.log
for wl in cfg3 cfg3_mpcc; do
  timeout 300 python bench.py --workload $wl --sets 8 --no-tight --latency-reps 0 --no-cpu-baseline --no-end-to-end --steps 20 --warmup 3 > $O/round4_${wl}_sets8.json 2> $O/round4_${wl}_sets8.err < /dev/null
done
timeout 300 python bench.py --workload jackal --no-tight --latency-reps 0 --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 > $O/round4_final_jackal.json 2> /dev/null < /dev/null
python3 - <<'PY'
import json,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out")
for f in ("round4_cfg3_sets8.json","round4_cfg3_mpcc_sets8.json","round4_final_jackal.json"):
    try:
        for l in open(os.path.join(O,f)):
            l=l.strip()
            if l.startswith("{"):
                j=json.loads(l); p=j.get("parity",{}); print(f, round(j["value"]), round(j["ms_per_step"],3), j["roofline"]["frac"], p.get("exit_code_mismatch"), p.get("parity_max_rel"), j["roofline"]["kernel"][:80])
    except Exception as e: print(f, "ERR", e)
PY
