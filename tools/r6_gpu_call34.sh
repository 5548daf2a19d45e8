#!/bin/bash
# jackal default on one-wave kernels at two lanes per stage (product rule): tests, saturated throughput, ticks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_compact2.py tests/test_gpu_gaussian.py tests/test_gpu_lds_poison.py tests/test_gpu_layout.py tests/test_cpp_optimize.py tests/test_gpu_tight.py -m gpu -q -x 2>&1 | tail -5
python bench.py --workload jackal --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print('jackal', round(d['value']), round(d['ms_per_step'],3), p.get('exit_code_mismatch'), p.get('sqp_iter_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'), d['roofline']['kernel'][:200])"
python tools/tick_shapes.py 100 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    if 'Gaussian' in d['shape']: print(d['shape'][:30], d['planners'], {m: (v['p50_ms'], v['kernel_ms'], v['exit_code_mismatch'] + v['sqp_iter_mismatch'] + v['ipm_iter_mismatch']) for m, v in d['by_mode'].items()})"
