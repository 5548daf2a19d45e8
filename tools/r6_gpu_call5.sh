#!/bin/bash
# round 6, GPU call 5: after pinning the structural Hh entries: modes 2 / 3 parity + phases + factorisation split; the tight + fixture + C++ variant tests
export TMPDIR=/tmp
mkdir -p gpurun_out build
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_tight.py tests/test_solve_fixtures.py tests/test_cpp_optimize.py tests/test_gpu_abi_contracts.py -x -q -m gpu > gpurun_out/r6_call5_tests.log 2>&1; tail -4 gpurun_out/r6_call5_tests.log
for m in 2 3; do python tools/profile_phases.py 64 $m; done > gpurun_out/r6_quad_phases3.jsonl 2> gpurun_out/r6_quad_phases3.err
cut -c1-420 gpurun_out/r6_quad_phases3.jsonl
python tools/make_newton_systems.py > gpurun_out/r6_newton.log 2>&1
build/scan_quad_bench 50 > gpurun_out/r6_scan_quad_bench2.json 2>&1; cut -c1-700 gpurun_out/r6_scan_quad_bench2.json
