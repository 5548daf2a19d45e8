#!/bin/bash
# round 6, GPU call 15: the predictor's forward elimination riding through the factorisation (modes 2 / 3): numerics on real Newton systems, parity, tick
export TMPDIR=/tmp
mkdir -p gpurun_out build
python tools/make_newton_systems.py > /dev/null 2>&1
build/scan_quad_bench 50 2>&1 | cut -c1-700 | tee gpurun_out/r6_scan_ridealong_bench.json
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_lds_poison.py tests/test_gpu_gaussian.py tests/test_cpp_optimize.py -x -q -m gpu > gpurun_out/r6_call15_tests.log 2>&1; tail -3 gpurun_out/r6_call15_tests.log
for m in 2 3; do python tools/profile_phases.py 64 $m; done 2>/dev/null | cut -c1-330
python tests/parity_sweep.py 24 > gpurun_out/round6_parity_sweep.jsonl 2>/dev/null; cat gpurun_out/round6_parity_sweep.jsonl
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "tests")
from mpc_planner_amd import scenes, solver
import oracle_lib as O
for nb, skw, dkw in ((64, dict(N=20, M=8), dict(N=20, S=5, n_lin=8, M=8)), (5, dict(N=20, M=8), dict(N=20, S=5, n_lin=8, M=8)), (64, dict(N=30, M=8, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1))):
    sc = scenes.make_scene(0, B=64, **skw)
    one = solver.BatchedSolver(solver.default_dims(**dkw), B_max=nb)
    for mode in (2, 3):
        if not one.set_latency_mode(mode): continue
        ts = []
        for i in range(210):
            t1 = time.perf_counter(); one.set_batch(sc["xinit"][:nb], sc["x0"][:nb], sc["params"][:nb]); one.solve(sync=False); b = one.select_best(); ts.append(time.perf_counter() - t1)
        g = one.get()
        xt, ut, o = O.solve_batch(O.problem(**dkw), sc["xinit"][:nb], sc["x0"][:nb].reshape(nb, -1), sc["params"][:nb].reshape(nb, -1))
        ok = o["exit_code"] == 1
        print(dkw["N"], "planners", nb, "mode", mode, "p50 ms", round(float(np.percentile(np.array(ts[10:]) * 1e3, 50)), 4), "int mism", int((g["exit_code"] != o["exit_code"]).sum() + (g["qp_iter_total"][ok] != o["qp_iter_total"][ok]).sum()), "max rel", float(np.abs(g["xtraj"][ok] - xt[ok]).max()))
    one.close()
PY
