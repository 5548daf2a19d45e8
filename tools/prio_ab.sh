#!/bin/bash
# A/B: s_setprio around the sequential (Riccati) phases of the compact one-wave kernels, and one PMC pass for the instruction cache.
#   tools/prio_ab.sh  (build, no GPU)  /  tools/prio_ab.sh measure  (GPU box)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
if [ "$1" != "measure" ]; then
  mkdir -p build/exp
  # variants: <sequential phases>_<rest of the interior-point iteration> (the linearisation is always 0); 0_0 = no priorities (rounds 1-4), 0_3 ~ the reverse
  for v in 1e-32 1e-28 1e-24 1e-20; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -disable-machine-licm -DTMPC_TU_COMPACT -DTMPC_MIRROR_TOL2=$v \
        -o build/exp/tmpc_solve_compact_prio$v.o mpc_planner_amd/csrc/tmpc_solve.hip &
  done
  wait
  for v in 1e-32 1e-28 1e-24 1e-20; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o build/exp/libtmpc_hip_prio$v.so build/exp/tmpc_solve_compact_prio$v.o \
        build/obj/tmpc_solve_fast.o build/obj/tmpc_solve_prof.o build/obj/tmpc_solve_cp2.o build/obj/tmpc_capi.o
  done
  ls -la build/exp/libtmpc_hip_prio*.so; exit 0
fi
export TMPDIR=/tmp
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 --gen-workers 1 --scene-cache /tmp/tmpc_bench_scenes"
for st in prio1e-32 prio1e-28 prio1e-24 prio1e-20; do
  export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_$st.so
  $BENCH 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['parity']
print(json.dumps({'variant':'$st','value':d['value'],'kernel_ms_avg':d['roofline']['kernel_ms_avg'],'parity':[p['exit_code_mismatch'],p['sqp_iter_mismatch'],p['ipm_iter_mismatch'],p['parity_max_rel']]}))"
done | tee gpurun_out/round5_h_mirror_tol_ab.jsonl
unset TMPC_HIP_LIBRARY
[ "$2" = pmc ] || exit 0
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" | head -30 > gpurun_out/round5_g_counters_avail.txt
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  d=gpurun_out/r5g_pmc; rm -rf $d
  rocprofv3 --pmc $c -d $d --output-format csv -- $BENCH --steps 3 --warmup 1 > /dev/null 2>&1
  python - "$c" <<'PY'
import csv, glob, sys, json
rows = []
for f in glob.glob('gpurun_out/r5g_pmc/**/*counter_collection.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if 'tmpc_solve_compact' in r.get('Kernel_Name', '')]
acc = {}
for r in rows:
    acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print(json.dumps({'counters': sys.argv[1], 'mean_per_launch': {k: sum(v) / len(v) for k, v in acc.items()}, 'launches': {k: len(v) for k, v in acc.items()}}))
PY
  rm -rf $d
done | tee gpurun_out/round5_g_icache_lds_pmc.jsonl
cat gpurun_out/round5_g_counters_avail.txt | head -20
