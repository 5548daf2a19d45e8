"""Kernel time per launch for the BASELINE config shapes (HIP events on the launch stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_planner_amd import scenes, solver
out = []
for name, kw, dims_kw, per_scene in (
        ("cfg1 MPCC+4 ellipsoids (no guidance), N=20", dict(N=20, M=4, B=64, guidance=False), dict(N=20, S=5, n_lin=0, M=4), 64),
        ("cfg2 T-MPC 8 obstacles, N=20", dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), 64),
        ("cfg4 T-MPC++ 12 obstacles, N=20", dict(N=20, M=12, B=63, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12), 64),
        ("reference default N=30, 8 obstacles (generic kernel)", dict(N=30, M=8, B=64), dict(N=30, S=5, n_lin=8, M=8), 64)):
    for n_scenes in (1, 8, 64):
        batch = scenes.make_batch(range(500, 500 + n_scenes), **kw)
        B = batch["xinit"].shape[0]
        s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=B)
        s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); s.solve()
        ms = s.time_solve(5); res = s.get(); s.close()
        out.append(dict(shape=name, B=B, kernel_ms=float(np.median(ms)), solves_per_s=B / (float(np.median(ms)) * 1e-3),
                        success=float((res["exit_code"] == 1).mean()), ipm_per_qp=float(res["qp_iter_total"].sum() / res["sqp_iter"].sum())))
        print(json.dumps(out[-1]), flush=True)
