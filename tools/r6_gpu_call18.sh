#!/bin/bash
# rotC = rounds 1-5 rotation + paired round-robin 4 x 4 sweep in the latency kernels, vs base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
OUT=r6_rotC_ab.jsonl bash tools/variants_ab.sh base rotC > /dev/null 2>&1; cat $O/r6_rotC_ab.jsonl
for v in base rotC base rotC; do
  TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so python tools/tick_shapes.py 100 2>/dev/null | tee -a $O/r6_rotC_tick_$v.jsonl | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    if d['shape'].startswith('cfg2') or d['shape'].startswith('jackalsim'):
        print('$v', d['shape'][:30], d['planners'], {m: (v['p50_ms'], v['kernel_ms'], v['exit_code_mismatch'] + v['sqp_iter_mismatch'] + v['ipm_iter_mismatch']) for m, v in d['by_mode'].items()})"
done
