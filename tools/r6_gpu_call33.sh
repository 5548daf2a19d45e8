#!/bin/bash
# experiment: mpc_planner_jackal's default (N = 30, 5 + 5 rows) on a ONE-wave compact kernel with two lanes per stage (lab switch TMPC_EXP_CP1W) vs the two-wave compact kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
export TMPC_HIP_LIBRARY=$R/mpc_planner_amd/libtmpc_hip_lab.so
for pass in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export TMPC_EXP_CP1W=1; else unset TMPC_EXP_CP1W; fi
  python bench.py --workload jackal --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 3 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print('cp1w=$v', round(d['value']), round(d['ms_per_step'],3), p.get('exit_code_mismatch'), p.get('sqp_iter_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'), d['roofline']['kernel'][:90])"
done; done
