#!/bin/bash
# re-take the cfg 5 lines (and the scenario tests) after a change to the scenario pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "scenario or shmpc or polygon or halfspace or support" > $O/cfg5_refresh_tests.log 2>&1 < /dev/null; tail -3 $O/cfg5_refresh_tests.log
timeout 300 python bench.py --workload cfg5 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg5.json 2> /dev/null < /dev/null
timeout 300 python bench.py --workload cfg5 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg5_mode2.json 2> /dev/null < /dev/null
timeout 300 python bench.py --workload cfg5 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg5_share8_mode2.json 2> /dev/null < /dev/null
python3 - <<'PY'
import json,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out")
for f in ("round4_final_cfg5.json","round4_final_cfg5_mode2.json","round4_final_cfg5_share8_mode2.json"):
    for l in open(os.path.join(O,f)):
        l=l.strip()
        if l.startswith("{"):
            j=json.loads(l); print(f, j["value"], j["ms_per_step"], j.get("parity",{}).get("exit_code_mismatch"), j.get("scenario_pipeline",{}).get("support_mean"))
PY
