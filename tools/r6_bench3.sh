#!/bin/bash
# the driver-form bench line three times on one box (run-to-run spread of the headline value)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
for i in 1 2 3; do python bench.py > $O/round6_final_bench_run$i.json 2> /dev/null; python -c "
import json
d=json.loads([l for l in open('$O/round6_final_bench_run$i.json') if l.startswith('{')][-1]); print($i, d['value'], d['ms_per_step'], d['roofline']['frac'], d['latency_b64']['p50_ms'], d['latency_b5']['p50_ms'], d.get('value_end_to_end'))"; done
