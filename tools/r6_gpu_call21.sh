#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; export TMPDIR=/tmp; mkdir -p $O
timeout 1200 python -m pytest tests/test_cpp_omp.py tests/test_cpp_optimize.py tests/test_cpp_solver.py tests/test_gpu_abi_contracts.py tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -4
python - 2>&1 <<'PY' | grep -v "^planner\|^$"
import sys, tempfile
sys.path.insert(0, "tests")
import test_cpp_omp as t
for planners, pp in ((7, True), (4, True)):
    out, kv = t.run_omp_ticks(tempfile.mkdtemp(), reps=100, planners=planners, tmpc_pp=pp)
    print("guidance planners", planners, "+ 1; rc", out.returncode); print(out.stdout)
PY
