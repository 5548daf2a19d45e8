#!/bin/bash
# four-wave linearisation: rows off the cost's wave (rows) vs rotC
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_lds_poison.py tests/test_gpu_gaussian.py -q -x 2>&1 | tail -4
for v in rotC rows rotC rows; do
  TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so python tools/tick_shapes.py 100 2>/dev/null | tee -a $O/r6_rows_tick_$v.jsonl | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$v', d['shape'][:30], d['planners'], {m: (v['p50_ms'], v['kernel_ms'], v['exit_code_mismatch'] + v['sqp_iter_mismatch'] + v['ipm_iter_mismatch']) for m, v in d['by_mode'].items() if m == 'mode_3'})"
done
for v in rotC rows; do TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v cfg5 mode 3', round(d['value']), round(d['ms_per_step'],4), d['parity'].get('exit_code_mismatch'), d['parity'].get('ipm_iter_mismatch'))"; done
