#!/bin/bash
# poly_clip with one division path (clip1) vs HEAD: polygon tests (Qhull-pinned, bitwise mirrors), saturated throughput, cfg 5 step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_clip1.so timeout 900 python -m pytest tests/test_polygon.py tests/test_gpu_shmpc_loop.py tests/test_sampler.py -m gpu -q -x 2>&1 | tail -3
for v in HEAD clip1 HEAD clip1; do
  [ $v = HEAD ] && unset TMPC_HIP_LIBRARY || export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so
  python tools/bench_polygon.py 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v saturated 2048', round(d['kernel_ms'],3))"
  python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v cfg5 mode 3', round(d['ms_per_step'],4), d['parity'].get('exit_code_mismatch'), d['scenario_pipeline'].get('support_mean'))"
done
