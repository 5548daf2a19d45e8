#!/bin/bash
# pinned transfer slabs of tick-size handles: full GPU suite, the C++ drop-in's tick, python ticks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python - 2>&1 <<'PY' | grep -v "^planner\|^$"
import sys, tempfile
sys.path.insert(0, "tests")
import test_cpp_omp as t
for planners, pp in ((7, True), (4, True)):
    out, kv = t.run_omp_ticks(tempfile.mkdtemp(), reps=100, planners=planners, tmpc_pp=pp)
    print("guidance planners", planners, "+ 1; rc", out.returncode); print(out.stdout)
PY
python tools/tick_shapes.py 100 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print(d['shape'][:30], d['planners'], {m: (v['p50_ms'], v['kernel_ms'], v['exit_code_mismatch'] + v['sqp_iter_mismatch'] + v['ipm_iter_mismatch']) for m, v in d['by_mode'].items()})"
