#!/bin/bash
# A/B of the libraries build/exp/libtmpc_hip_<name>.so (tools/build_compact_variants.sh) on the driver-form workload, interleaved, two passes
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; O=gpurun_out/${OUT:-variants_ab.jsonl}; : > $O
for pass in 1 2; do
for name in "$@"; do
  export TMPC_HIP_LIBRARY=$PWD/build/exp/libtmpc_hip_$name.so
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 0 --index-check-sets 0 --scene-cache /tmp/sc.npz 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'variant':'$name','pass':$pass,'value':d['value'],'kernel_ms_avg':d['roofline']['kernel_ms_avg']}))" >> $O
done; done
cat $O
