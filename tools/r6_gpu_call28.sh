#!/bin/bash
# tmpc_scenario_halfspaces: seeds of the second filter round by table (polytab) vs the pairwise comparison (product library of HEAD)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_polytab.so timeout 900 python -m pytest tests/test_gpu_scenario.py tests/test_gpu_scenario_pipeline.py tests/test_scenario_polygon.py -m gpu -q -x 2>&1 | tail -4
for v in HEAD polytab; do
  [ $v = HEAD ] && unset TMPC_HIP_LIBRARY || export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_polytab.so
  for m in 3 0; do python bench.py --workload cfg5 --latency-mode $m --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print('$v mode $m', round(d['value']), round(d['ms_per_step'],4), p.get('exit_code_mismatch'), p.get('ipm_iter_mismatch'), d['scenario_pipeline'].get('support_mean'), d['scenario_pipeline'].get('empty_polygon_stages'))"; done
done
python tools/bench_polygon.py 8 2>/dev/null | tail -4
TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_polytab.so python tools/bench_polygon.py 8 2>/dev/null | tail -4
