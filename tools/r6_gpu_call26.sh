#!/bin/bash
# cfg 5 (SH-MPC) step: per-kernel durations of the pipeline (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
rm -rf $O/cfg5_trace
timeout 600 rocprofv3 --kernel-trace --stats -d $O/cfg5_trace --output-format csv -- python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 --parity-check 0 --index-check-sets 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/cfg5_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', r['Percentage'])
PY
rm -rf $O/cfg5_trace
