#!/bin/bash
# (the TMPC_* kernel-selection switches exist in the lab build of the library only: round 6)
export TMPC_HIP_LIBRARY=${TMPC_HIP_LIBRARY:-${GRAFT_REPO_ROOT:-/root/repo}/mpc_planner_amd/libtmpc_hip_lab.so}
# Throughput of the compact kernel against residency (workgroups per CU), TMPC_COMPACT_PER_CU lab switch.
mkdir -p gpurun_out
out=gpurun_out/round5_l_per_cu.jsonl
: > $out
for p in 4 5 6 7 8; do
  TMPC_COMPACT_PER_CU=$p timeout 300 python bench.py --steps 10 --warmup 2 --no-end-to-end --no-tight --no-cpu-baseline --scene-cache /tmp/sc.npz --latency-reps 5 \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'per_cu': $p, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline'].get('kernel_ms_avg')}))" >> $out
done
cat $out
