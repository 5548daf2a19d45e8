#!/bin/bash
# A/B of the Hh blocks' stage-stride padding in the compact layout (Dims::hpad): TMPC_EXP_HPAD=0 is the bare stride (28 doubles)
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; O=gpurun_out/round5_q_hpad_ab.jsonl; : > $O
run() { # name, env value ('' = padded when the residency allows), bench args...
  local name=$1 v=$2; shift 2
  ( if [ -n "$v" ]; then export TMPC_EXP_HPAD=$v; fi
    python bench.py "$@" --no-cpu-baseline --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 --latency-reps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'case':'$name','hpad':'${v:-auto}','value':d['value'],'ms_per_step':d['ms_per_step'],'lds':d['config'].get('kernel','')[:0],'parity':[ (d.get('parity') or {}).get(k) for k in ('exit_code_mismatch','sqp_iter_mismatch','ipm_iter_mismatch','parity_max_rel')]}))" >> $O )
}
for pass in 1 2; do
for v in 0 ""; do
  run cfg2 "$v" --steps 12 --warmup 3 --scene-cache /tmp/sc2.npz
  run cfg3_sets8 "$v" --workload cfg3 --sets 8 --steps 20 --warmup 3
  run jackal "$v" --workload jackal --steps 8 --warmup 2
done; done
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_iterations.py -m gpu -x -q 2>&1 | tail -3
