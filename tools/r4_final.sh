#!/bin/bash
# round 4, final measurement set (one MI355X): the driver-form bench line, rocprofv3 kernel stats + PMC of the same command, the other
# BASELINE configs (as named; small sets also with the tick variant), in-kernel phase profile, full GPU test suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
( time python bench.py ) > $O/round4_final_bench.json 2> $O/round4_final_bench.err
python tools/collect_profiles.py round4_final > $O/round4_final_collect.log 2>&1
for wl in cfg3 cfg3_mpcc cfg4 cfg5 jackal; do
  timeout 400 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_${wl}.json 2> /dev/null
done
timeout 400 python bench.py --workload cfg4 --share-of 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg4_share8.json 2> /dev/null
timeout 400 python bench.py --workload cfg4 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg4_share8_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg5_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round4_final_cfg5_share8_mode2.json 2> /dev/null
python tools/profile_phases.py 64 0 > $O/round4_final_phases.jsonl 2>/dev/null; python tools/profile_phases.py 64 2 >> $O/round4_final_phases.jsonl 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/round4_final_gpu_suite.log 2>&1; tail -3 $O/round4_final_gpu_suite.log
tail -4 $O/round4_final_bench.err
