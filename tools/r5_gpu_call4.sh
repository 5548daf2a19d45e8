#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -s > $O/r5_call4_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/r5_call4_pytest.log | cut -c1-400; grep "\[parity\]" $O/r5_call4_pytest.log | tail -1
timeout 300 python bench.py --workload jackal --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_d_jackal_tuned.json 2> /dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/round5_d_jackal_tuned.json') if l.startswith('{')][-1])
print("jackal value", d['value'], "ms", d['ms_per_step'], d['roofline']['kernel'], {k: v for k, v in d['parity'].items() if k not in ('best_index','sample','against')})
PY
