#!/bin/bash
# A/B of the AMDGPU instruction-scheduling strategy for the compact one-wave kernels (the cfg 2 bench kernel): does another scheduler fill more of the
# ~30 % of issue slots in which both resident waves stall?   tools/sched_ab.sh  (build, no GPU)  /  tools/sched_ab.sh measure  (GPU box)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
STRATS="max-ilp max-memory-clause iterative-ilp"
if [ "$1" != "measure" ]; then
  mkdir -p build/exp
  for st in $STRATS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -disable-machine-licm -DTMPC_TU_COMPACT -mllvm -amdgpu-sched-strategy=$st \
        -o build/exp/tmpc_solve_compact_$st.o mpc_planner_amd/csrc/tmpc_solve.hip &
  done
  wait
  for st in $STRATS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o build/exp/libtmpc_hip_sched_$st.so build/exp/tmpc_solve_compact_$st.o \
        build/obj/tmpc_solve_fast.o build/obj/tmpc_solve_prof.o build/obj/tmpc_solve_cp2.o build/obj/tmpc_capi.o
  done
  ls -la build/exp/libtmpc_hip_sched_*.so; exit 0
fi
export TMPDIR=/tmp
for st in default $STRATS; do
  if [ "$st" != default ]; then export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_sched_$st.so; else unset TMPC_HIP_LIBRARY; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --latency-reps 0 --no-tight --no-end-to-end --parity-check 64 --index-check-sets 0 --gen-workers 1 --scene-cache /tmp/tmpc_bench_scenes 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['parity']
print(json.dumps({'strategy':'$st','value':d['value'],'kernel_ms_avg':d['roofline']['kernel_ms_avg'],'parity':[p['exit_code_mismatch'],p['sqp_iter_mismatch'],p['ipm_iter_mismatch'],p['parity_max_rel']]}))"
done | tee gpurun_out/round5_f_sched_strategy_ab.jsonl
