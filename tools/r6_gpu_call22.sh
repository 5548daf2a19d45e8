#!/bin/bash
# quad pre-reduction of the stage sums in the two-wave kernels at four lanes per stage (cfg 3, jackal default): base2 vs qsum4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O; : > $O/r6_qsum4_ab.jsonl
for pass in 1 2; do for v in base2 qsum4; do
  export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so
  for wl in "jackal" "cfg3" "cfg3_mpcc" "cfg3 --sets 8" "cfg3_mpcc --latency-mode 2"; do
    timeout 300 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 30 --warmup 5 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print(json.dumps({'variant':'$v','pass':$pass,'workload':'$wl','value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'mismatch':[p.get('exit_code_mismatch'),p.get('sqp_iter_mismatch'),p.get('ipm_iter_mismatch')],'rel':p.get('parity_max_rel')}))" | tee -a $O/r6_qsum4_ab.jsonl
  done
done; done
