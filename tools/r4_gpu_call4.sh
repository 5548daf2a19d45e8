#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_ca.py -m gpu -x -q > $O/r4_c4_tests.log 2>&1; tail -4 $O/r4_c4_tests.log
timeout 900 python bench.py --no-tight --latency-reps 0 --no-cpu-baseline --index-check-sets 64 > $O/r4_c4_bench.json 2> $O/r4_c4_bench.err; tail -c 600 $O/r4_c4_bench.err
