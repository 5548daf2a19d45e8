#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
export TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_polytab2.so
timeout 900 python -m pytest $(grep -ln "scenario_halfspaces" tests/*.py) -m gpu -q -x 2>&1 | tail -4
rm -rf $O/cfg5_trace
timeout 600 rocprofv3 --kernel-trace -d $O/cfg5_trace --output-format csv -- python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 5 --parity-check 0 --index-check-sets 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/cfg5_trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'halfspaces' in r['Kernel_Name']: acc[r['Grid_Size_X']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in acc.items(): v.sort(); print('halfspaces grid', k, len(v), 'p50', v[len(v)//2], 'min', v[0], 'max', v[-1])
PY
rm -rf $O/cfg5_trace
TMPC_HIP_LIBRARY=$R/build/exp/libtmpc_hip_polytab2.so python tools/bench_polygon.py 8 2>/dev/null | tail -4
