#!/bin/bash
# cfg 5 step: the two passes of tmpc_scenario_halfspaces separately (per-dispatch kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
rm -rf $O/cfg5_trace
timeout 600 rocprofv3 --kernel-trace -d $O/cfg5_trace --output-format csv -- python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 5 --parity-check 0 --index-check-sets 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/cfg5_trace/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(list(rows[0].keys()))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    prev_end = None
    for r in rows:
        if 'tmpc' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:46], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('LDS_Block_Size') or r.get('LDS_Block_Size_v'))
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        gap = (int(r['Start_Timestamp']) - prev_end) / 1e3 if prev_end else 0
        acc[key].append((dur, gap)); prev_end = int(r['End_Timestamp'])
for k, v in acc.items():
    d = sorted(x[0] for x in v); g = sorted(x[1] for x in v)
    print(k, len(v), 'dur p50', d[len(d)//2], 'us; gap before p50', g[len(g)//2], 'us')
PY
rm -rf $O/cfg5_trace
