#!/bin/bash
# one-wave kernels at two lanes per stage: pair pre-reduction of the stage sums (HEAD) vs base5
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compact2.py tests/test_gpu_gaussian.py -m gpu -q -x 2>&1 | tail -3
for pass in 1 2; do for v in base5 HEAD; do
  [ $v = HEAD ] && unset TMPC_HIP_LIBRARY || export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so
  python bench.py --workload jackal --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 3 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print('$v jackal', round(d['value']), round(d['ms_per_step'],3), p.get('exit_code_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'))"
  python tools/n30_throughput.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v n30 8+8', round(d['solves_per_s']), d['parity_128'])"
done; done
