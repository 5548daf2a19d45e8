#!/bin/bash
# Full library variant (every translation unit recompiled with extra flags): build/exp/libtmpc_hip_<name>.so, lab C-ABI unit (TMPC_* switches read).
#   tools/build_full_variant.sh <name> [-DTMPC_EXP_...]
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/build/exp/full_$name; mkdir -p $D
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage"
for tu in FAST COMPACT PROF CP2 SQRT QUAD QUADW; do
  $C -DTMPC_TU_$tu "$@" -o $D/solve_$tu.o $R/mpc_planner_amd/csrc/tmpc_solve.hip 2> $D/solve_$tu.log &
done
$C -DTMPC_LAB_SWITCHES "$@" -o $D/capi.o $R/mpc_planner_amd/csrc/tmpc_capi.hip 2> $D/capi.log &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $R/build/exp/libtmpc_hip_$name.so $D/*.o
echo "$name: $(cat $D/*.log | grep -c 'ScratchSize \[bytes/lane\]: [1-9]') kernels with scratch"
