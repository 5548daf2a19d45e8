#!/bin/bash
# tmpc_scenario_halfspaces after the round-6 changes (table seeds for long lists, G by cost): tests, saturated polygon throughput, cfg 5 step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest $(grep -ln "scenario_halfspaces" tests/*.py) -m gpu -q -x 2>&1 | tail -3
python tools/bench_polygon.py 8 2>/dev/null | tail -4
for m in 3 2 0; do python bench.py --workload cfg5 --latency-mode $m --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print('cfg5 mode $m', round(d['value']), round(d['ms_per_step'],4), p.get('exit_code_mismatch'), p.get('ipm_iter_mismatch'), d['scenario_pipeline'].get('support_mean'))"; done
