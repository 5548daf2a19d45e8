#!/bin/bash
# round 4, GPU call 2: new tests (CA-MPC, slot reuse, SH-MPC loop after the scenes move) + the whole GPU suite, bench lines of cfg3 (CA), cfg3_mpcc, cfg5 (pipeline in the step)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_ca.py tests/test_gpu_shmpc_loop.py tests/test_cpp_solver_api.py tests/test_cpp_optimize.py -m gpu -x -q > $O/r4_c2_newtests.log 2>&1
tail -5 $O/r4_c2_newtests.log
for wl in cfg3 cfg3_mpcc cfg5; do
  timeout 400 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/r4_${wl}_bench.json 2> $O/r4_${wl}_bench.err
  tail -c 400 $O/r4_${wl}_bench.err
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_c2_fullsuite.log 2>&1
tail -5 $O/r4_c2_fullsuite.log
