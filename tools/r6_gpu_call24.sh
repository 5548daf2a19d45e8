#!/bin/bash
# six lanes per stage (two-wave latency kernels, N <= 21): stage sums pre-reduced per triple (tsum6) vs base4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O; : > $O/r6_tsum6_ab.jsonl
for pass in 1 2; do for v in base4 tsum6; do
  export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$v.so
  for wl in "cfg4 --share-of 8 --latency-mode 2" "cfg5 --latency-mode 2" "cfg4 --share-of 8 --latency-mode 1"; do
    timeout 300 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 --index-check-sets 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('parity') or {}
print(json.dumps({'variant':'$v','pass':$pass,'workload':'$wl','value':round(d['value']),'ms_per_step':round(d['ms_per_step'],4),'mismatch':[p.get('exit_code_mismatch'),p.get('sqp_iter_mismatch'),p.get('ipm_iter_mismatch')],'rel':p.get('parity_max_rel')}))" | tee -a $O/r6_tsum6_ab.jsonl
  done
  python tools/tick_shapes.py 100 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    if d['shape'].startswith('cfg2'):
        print(json.dumps({'variant':'$v','pass':$pass,'tick':d['shape'][:12],'planners':d['planners'],'by_mode':{m:[x['p50_ms'],x['kernel_ms'],x['exit_code_mismatch']+x['sqp_iter_mismatch']+x['ipm_iter_mismatch']] for m,x in d['by_mode'].items() if m in ('mode_1','mode_2')}}))" | tee -a $O/r6_tsum6_ab.jsonl
done; done
