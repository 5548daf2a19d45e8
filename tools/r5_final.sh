#!/bin/bash
# round 5, final measurement set (one MI355X): the driver-form bench line, rocprofv3 kernel stats + PMC of the same command, the other
# BASELINE configs (as named; small sets also with the tick variants), in-kernel phase profile, full GPU test suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp
( time python bench.py ) > $O/round5_final_bench.json 2> $O/round5_final_bench.err
python tools/collect_profiles.py round5_final > $O/round5_final_collect.log 2>&1
for wl in cfg3 cfg3_mpcc cfg4 cfg5 jackal; do
  timeout 400 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_final_${wl}.json 2> /dev/null
done
timeout 400 python bench.py --workload cfg3 --sets 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 3 > $O/round5_final_cfg3_sets8.json 2> /dev/null
timeout 400 python bench.py --workload cfg4 --share-of 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_final_cfg4_share8.json 2> /dev/null
timeout 400 python bench.py --workload cfg4 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_final_cfg4_share8_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_final_cfg5_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round5_final_cfg5_share8_mode2.json 2> /dev/null
python tools/profile_phases.py 64 0 > $O/round5_final_phases.jsonl 2>/dev/null; python tools/profile_phases.py 64 1 >> $O/round5_final_phases.jsonl 2>/dev/null; python tools/profile_phases.py 64 2 >> $O/round5_final_phases.jsonl 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/round5_final_gpu_suite.log 2>&1; tail -3 $O/round5_final_gpu_suite.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/round5_final_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if 'value' in d:
        p = d.get('parity') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'frac', round(d['roofline']['frac'], 4), 'parity', p.get('exit_code_mismatch'), p.get('sqp_iter_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'))
d = json.loads([l for l in open('gpurun_out/round5_final_bench.json') if l.startswith('{')][-1])
print('e2e', d.get('value_end_to_end'), 'tight', d.get('value_qp_tol_1e_9'), 'lat', d['latency_b64']['p50_ms'], d['latency_b64']['two_wave_riccati']['p50_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
tail -4 $O/round5_final_bench.err
