#!/bin/bash
# cfg 5 pipeline: per-kernel durations with different first-pass candidate-list sizes of tmpc_scenario_halfspaces (TMPC_POLY_LIST_CAP)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
for cap in 512 1024 2048; do
  TMPC_POLY_LIST_CAP=$cap timeout 300 bash $R/tools/profile_workload.sh polycap_$cap --workload cfg5 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --no-end-to-end --steps 50 --warmup 5 > /dev/null 2>&1 < /dev/null
  echo "cap $cap"; grep -h "halfspaces\|solve_fast" $O/polycap_${cap}_kernel_stats.csv 2>/dev/null | awk -F'",' '{print "   ", substr($1,1,50), $2, $4, $6, $7}' | tr -d '"'
  python3 -c "
import json,sys
for l in open('$O/polycap_${cap}_bench.json'):
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('    ms_per_step', j['ms_per_step'])
" 2>/dev/null
done
