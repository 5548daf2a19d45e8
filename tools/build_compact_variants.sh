#!/bin/bash
# A/B libraries that differ only in one solve translation unit (TU=TMPC_TU_COMPACT by default, or TMPC_TU_CP2): build/exp/libtmpc_hip_<name>.so for every "name:flags" argument
# (the other objects are the product build's, the C-ABI unit the lab one -- it reads the TMPC_* switches --: run __graft_entry__.build() first).  Usage: tools/build_compact_variants.sh "u1r0:-DTMPC_FACTOR_UNROLL=1 -DTMPC_ROT_BRANCH=0" ...
R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/build/exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -mllvm -disable-machine-licm -D${TU:-TMPC_TU_COMPACT} $flags -Rpass-analysis=kernel-resource-usage \
      -o $R/build/exp/compact_$name.o $R/mpc_planner_amd/csrc/tmpc_solve.hip 2> $R/build/exp/compact_$name.log \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $R/build/exp/libtmpc_hip_$name.so $R/build/obj/tmpc_solve_fast.o $([ "${TU:-TMPC_TU_COMPACT}" = TMPC_TU_COMPACT ] && echo $R/build/exp/compact_$name.o || echo $R/build/obj/tmpc_solve_compact.o) \
         $R/build/obj/tmpc_solve_prof.o $([ "${TU:-TMPC_TU_COMPACT}" = TMPC_TU_CP2 ] && echo $R/build/exp/compact_$name.o || echo $R/build/obj/tmpc_solve_cp2.o) $R/build/obj/tmpc_solve_sqrt.o $R/build/obj/tmpc_solve_quad.o $R/build/obj/tmpc_solve_quadw.o $R/build/obj/tmpc_capi_lab.o \
    && echo "$name: $(grep -c 'ScratchSize \[bytes/lane\]: [1-9]' $R/build/exp/compact_$name.log) kernels with scratch" ) &
done
wait
