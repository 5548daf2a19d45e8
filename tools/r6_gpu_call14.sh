#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=r6_levers_ab6.jsonl bash tools/variants_ab.sh rhscopy rhsdiff > /dev/null 2>&1
cat gpurun_out/r6_levers_ab6.jsonl
TMPC_HIP_LIBRARY=$PWD/build/exp/libtmpc_hip_rhsdiff.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 1024 --index-check-sets 0 --scene-cache /tmp/sc.npz 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('rhsdiff parity', {k: d['parity'][k] for k in ('trajectories','exit_code_mismatch','sqp_iter_mismatch','ipm_iter_mismatch','parity_max_rel')})"
