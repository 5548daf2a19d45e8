#!/bin/bash
# (the TMPC_* kernel-selection switches exist in the lab build of the library only: round 6)
export TMPC_HIP_LIBRARY=${TMPC_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/mpc_planner_amd/libtmpc_hip_lab.so}
# A/B of the row Jacobians' stage-stride padding in LDS (Dims::dpad, tmpc_capi.hip pick_d_pad): TMPC_EXP_DPAD=0 is the bare stride
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; O=gpurun_out/round5_p_dpad_ab.jsonl; : > $O
run() { # name, env value ('' = the model's choice), bench args...
  local name=$1 v=$2; shift 2
  ( if [ -n "$v" ]; then export TMPC_EXP_DPAD=$v; fi
    python bench.py "$@" --no-cpu-baseline --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
lat=d.get('latency_b64') or {}
print(json.dumps({'case':'$name','dpad':'${v:-model}','value':d['value'],'ms_per_step':d['ms_per_step'],'tick_p50_ms':lat.get('p50_ms'),'tick_mode1_p50_ms':(lat.get('two_wave_riccati') or {}).get('p50_ms'),'parity':[ (d.get('parity') or {}).get(k) for k in ('exit_code_mismatch','sqp_iter_mismatch','ipm_iter_mismatch','parity_max_rel')]}))" >> $O )
}
for pass in 1 2; do
for v in 0 ""; do
  run cfg2 "$v" --steps 12 --warmup 3 --latency-reps 100 --scene-cache /tmp/sc2.npz
  run cfg4 "$v" --workload cfg4 --steps 50 --warmup 5 --latency-reps 0
  run cfg4_share8 "$v" --workload cfg4 --share-of 8 --steps 50 --warmup 5 --latency-reps 0
  run cfg3 "$v" --workload cfg3 --steps 50 --warmup 5 --latency-reps 0
  run cfg5 "$v" --workload cfg5 --steps 50 --warmup 5 --latency-reps 0
done; done
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact2.py tests/test_gpu_iterations.py tests/test_gpu_latency_mode.py -m gpu -x -q 2>&1 | tail -3
