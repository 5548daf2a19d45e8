#!/bin/bash
# after a layout change: the saturated workloads of every compact kernel family + the solve parity tests
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; O=gpurun_out/${OUT:-layout_check.jsonl}; : > $O
run() { local name=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'case':'$name','value':d['value'],'ms_per_step':d['ms_per_step'],'parity':[ (d.get('parity') or {}).get(k) for k in ('exit_code_mismatch','sqp_iter_mismatch','ipm_iter_mismatch','parity_max_rel')]}))" >> $O
}
for pass in 1 2; do
  run cfg2 --steps 12 --warmup 3 --scene-cache /tmp/sc2.npz
  run cfg4 --workload cfg4 --steps 50 --warmup 5
  run cfg3 --workload cfg3 --steps 50 --warmup 5
  run cfg3_sets8 --workload cfg3 --sets 8 --steps 20 --warmup 3
  run jackal --workload jackal --steps 20 --warmup 3
  run cfg5 --workload cfg5 --steps 50 --warmup 5
done
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_iterations.py -m gpu -x -q 2>&1 | tail -3
