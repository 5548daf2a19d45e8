#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python bench.py --no-tight --latency-reps 20 --no-cpu-baseline --index-check-sets 16 --no-end-to-end > $O/r4_c5_bench.json 2> $O/r4_c5_bench.err; tail -c 300 $O/r4_c5_bench.err
for wl in cfg4 cfg5 cfg3; do timeout 400 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/r4_c5_${wl}.json 2> $O/r4_c5_${wl}.err; done
timeout 400 python bench.py --workload cfg4 --share-of 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/r4_c5_cfg4_share8.json 2> /dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_c5_fullsuite.log 2>&1; tail -5 $O/r4_c5_fullsuite.log
