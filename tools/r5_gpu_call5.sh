#!/bin/bash
# round 5, GPU call 2: the Riccati recursion without the five state pivots -- GPU suite, bench line, phase profile, the other configs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/r5_call5_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/r5_call5_pytest.log | cut -c1-300; grep "\[parity\]" $O/r5_call5_pytest.log | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_e_bench.json 2> $O/r5_call5_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/round5_e_bench.json') if l.startswith('{')][-1])
print("value", d['value'], "ms", d['ms_per_step'], "frac", d['roofline']['frac'], "e2e", d.get('value_end_to_end'), "tight", d.get('value_qp_tol_1e_9'))
print("parity", {k: v for k, v in d['parity'].items() if k != 'best_index'}, d['parity']['best_index']['true_mismatches'], d['parity']['best_index']['best_index_mismatch_vs_oracle'])
print("tight parity", d['qp_tol_1e_9']['parity'])
print("e2e", json.dumps(d['end_to_end']['vs_resident_step'])); print("proj", json.dumps(d['end_to_end']['projection_exercise']))
print("lat", d['latency_b64']['p50_ms'], d['latency_b64']['kernel_ms_b64'], d['latency_b64']['two_wave_riccati']['p50_ms'], d['latency_b64']['two_wave_riccati']['kernel_ms_b64'])
PY
python tools/profile_phases.py 64 0 > $O/round5_e_phases.jsonl 2>/dev/null; python tools/profile_phases.py 64 1 >> $O/round5_e_phases.jsonl 2>/dev/null; cat $O/round5_e_phases.jsonl | cut -c1-400
for wl in cfg4; do
  timeout 300 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_e_${wl}.json 2> /dev/null
  python - "$wl" <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/round5_e_{sys.argv[1]}.json') if l.startswith('{')][-1])
print(sys.argv[1], "value", d['value'], "ms", d['ms_per_step'], {k: v for k, v in d['parity'].items() if k not in ('best_index','sample','against')})
PY
done
