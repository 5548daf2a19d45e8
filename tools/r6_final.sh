#!/bin/bash
# round 6, final measurement set (one MI355X): the driver-form bench line, rocprofv3 kernel stats + PMC of the same command, the other
# BASELINE configs (as named; small sets also with the tick variants), in-kernel phase profile, the un-patched OpenMP drop-in, full GPU test suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
python tools/collect_profiles.py round6_final > $O/round6_final_collect.log 2>&1
cp $O/round6_final_pmc.json $O/round6_final_rocprof_summary.json profiles/      # (on the box: the bench line below reports roofline.traffic from the counters of THIS build)
( time python bench.py ) > $O/round6_final_bench.json 2> $O/round6_final_bench.err
for i in 2 3; do python bench.py 2> /dev/null | grep '^{' | tail -1 >> $O/round6_final_bench_repeats.jsonl; done
for wl in cfg3 cfg3_mpcc cfg4 cfg5 jackal; do
  timeout 400 python bench.py --workload $wl --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_${wl}.json 2> /dev/null
done
timeout 400 python bench.py --workload cfg3 --sets 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 20 --warmup 3 > $O/round6_final_cfg3_sets8.json 2> /dev/null
timeout 400 python bench.py --workload cfg4 --share-of 8 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_cfg4_share8.json 2> /dev/null
timeout 400 python bench.py --workload cfg4 --share-of 8 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_cfg4_share8_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --latency-mode 2 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_cfg5_mode2.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_cfg5_mode3.json 2> /dev/null
timeout 400 python bench.py --workload cfg5 --share-of 8 --latency-mode 3 --no-tight --latency-reps 0 --no-cpu-baseline --steps 50 --warmup 5 > $O/round6_final_cfg5_share8_mode3.json 2> /dev/null
timeout 600 python tools/tick_shapes.py 100 > $O/round6_final_tick_shapes.jsonl 2> /dev/null      # ticks at the horizon the reference ships (N = 30), every variant, with parity
: > $O/round6_final_phases.jsonl; for m in 0 1 2 3; do python tools/profile_phases.py 64 $m >> $O/round6_final_phases.jsonl 2>/dev/null; done
# the un-patched drop-in: eight Solvers on eight OpenMP threads vs one solveBatch launch (tests/cpp/test_omp_solvers.cpp)
python - > $O/round6_final_omp_dropin.log 2>&1 <<'PY'
import sys, tempfile
sys.path.insert(0, "tests")
import test_cpp_omp as t
for planners, pp, horizon in ((7, True, 20), (4, True, 20), (4, True, 30)):
    out, kv = t.run_omp_ticks(tempfile.mkdtemp(), reps=100, planners=planners, tmpc_pp=pp, horizon=horizon)
    print("guidance planners", planners, "+ the non-guided planner; horizon N =", horizon, "; return code", out.returncode)
    print(out.stdout)
PY
cat $O/round6_final_omp_dropin.log | grep -v "^planner\|^$"
timeout 2400 python -m pytest tests -m gpu -x -q > $O/round6_final_gpu_suite.log 2>&1; tail -3 $O/round6_final_gpu_suite.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/round6_final_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if 'value' in d:
        p = d.get('parity') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'frac', round(d['roofline']['frac'], 4), 'parity', p.get('exit_code_mismatch'), p.get('sqp_iter_mismatch'), p.get('ipm_iter_mismatch'), p.get('parity_max_rel'))
d = json.loads([l for l in open('gpurun_out/round6_final_bench.json') if l.startswith('{')][-1])
print('e2e', d.get('value_end_to_end'), 'tight 1e-8', d.get('value_qp_tol_1e_8'), d['qp_tol_1e_8']['parity'], 'beyond 1e-9', d['qp_tol_1e_8']['beyond_the_noise_floor_1e_9']['parity'])
for l in open('gpurun_out/round6_final_tick_shapes.jsonl'):
    t = json.loads(l)
    print('tick', t['shape'][:60], t['planners'], {m: (v['p50_ms'], v['exit_code_mismatch'] + v['sqp_iter_mismatch'] + v['ipm_iter_mismatch']) for m, v in t['by_mode'].items()})
print('lat64', d['latency_b64']['p50_ms'], d['latency_b64']['fastest_mode'], 'lat5', d['latency_b5']['p50_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'best_index', d['parity'].get('best_index'))
PY
tail -4 $O/round6_final_bench.err
