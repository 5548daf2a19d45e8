#!/bin/bash
# round 6, GPU call 4: four-wave kernel after the quad pre-reduction; cycle split of the four-wave factorisation on real Newton systems
export TMPDIR=/tmp
mkdir -p gpurun_out build
timeout 600 python -m pytest tests/test_gpu_quad.py -x -q > gpurun_out/r6_quad_tests.log 2>&1; tail -2 gpurun_out/r6_quad_tests.log
for m in 2 3; do python tools/profile_phases.py 64 $m; done > gpurun_out/r6_quad_phases2.jsonl 2> gpurun_out/r6_quad_phases2.err
cut -c1-420 gpurun_out/r6_quad_phases2.jsonl
python tools/make_newton_systems.py > gpurun_out/r6_newton.log 2>&1
build/scan_quad_bench 50 > gpurun_out/r6_scan_quad_bench.json 2>&1; cat gpurun_out/r6_scan_quad_bench.json
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "tests")
from mpc_planner_amd import scenes, solver
sc = scenes.make_scene(5, N=20, M=8, B=64)
for nb in (64, 5):
    one = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8), B_max=nb)
    for mode in (2, 3):
        one.set_latency_mode(mode)
        one.set_batch(sc["xinit"][:nb], sc["x0"][:nb], sc["params"][:nb]); one.solve()
        one.get_timings(); one.enable_timing(64)
        for _ in range(40): one.solve(sync=False)
        one.synchronize()
        print("B", nb, "mode", mode, "kernel ms median", float(np.median(one.get_timings())))
    one.close()
PY
