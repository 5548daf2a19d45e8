#!/bin/bash
# round 6, GPU call 12: full GPU suite (lab-library tests, Gaussian rows on the latency / compact kernels, the Riccati changes); jackal default saturated + its tick
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_suite3.log 2>&1
tail -6 gpurun_out/r6_gpu_suite3.log
timeout 900 python bench.py --workload jackal --steps 8 --warmup 2 --no-cpu-baseline --latency-reps 0 > gpurun_out/r6_jackal_bench.json 2> gpurun_out/r6_jackal_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_jackal_bench.json") if l.startswith("{")][-1])
print("jackal value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"][:60], "parity", {k: d["parity"][k] for k in ("exit_code_mismatch", "sqp_iter_mismatch", "ipm_iter_mismatch", "parity_max_rel")})
PY
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "tests")
from mpc_planner_amd import scenes, solver
sc = scenes.make_scene(5, N=30, M=5, S=3, chance=True, B=64)
for nb in (64, 5):
    one = solver.BatchedSolver(solver.default_dims(N=30, S=3, n_lin=5, M=5, row_model=1), B_max=nb)
    for mode in (0, 2):
        ok = one.set_latency_mode(mode) if mode else True
        ts = []
        for i in range(110):
            t1 = time.perf_counter(); one.set_batch(sc["xinit"][:nb], sc["x0"][:nb], sc["params"][:nb]); one.solve(sync=False); b = one.select_best(); ts.append(time.perf_counter() - t1)
        print("jackal default tick, planners", nb, "mode", mode, "accepted", ok, "p50 ms", round(float(np.percentile(np.array(ts[10:]) * 1e3, 50)), 4))
    one.close()
PY
