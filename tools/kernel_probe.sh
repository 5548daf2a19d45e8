#!/bin/bash
# Compile ONE instantiation of a solve kernel and print its resource usage (registers, scratch, occupancy).
#   tools/kernel_probe.sh 8,8,3,64,false [extra hipcc flags...]      fast kernel      -> /tmp/probe.{o,log}
#   COMPACT=1 tools/kernel_probe.sh 8,8,3,false                        compact kernel
set -e
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT/mpc_planner_amd/csrc/tmpc_solve.hip}
OUT=${OUT:-/tmp/probe}
DEF=TMPC_SINGLE_KERNEL; [ -n "$COMPACT" ] && DEF=TMPC_SINGLE_COMPACT;
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${EMIT:--c} -mllvm -disable-machine-licm "-D$DEF=$ARGS" \
    -Rpass-analysis=kernel-resource-usage --cuda-device-only "$@" -o $OUT.${EXT:-o} $SRC 2> $OUT.log
python3 $ROOT/tools/kernel_resources.py $OUT.log
