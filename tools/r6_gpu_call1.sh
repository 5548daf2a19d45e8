#!/bin/bash
# round 6, GPU call 1: did anything break (GPU suite), the un-patched OpenMP drop-in, the square-root Riccati instantiations, the tight-tolerance flips
export TMPDIR=/tmp
mkdir -p gpurun_out
python - > gpurun_out/r6_sqrt_check.log 2>&1 <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "tests")
from mpc_planner_amd import scenes, solver
import oracle_lib as O
def run(name, dims_kw, scene_kw, B, orc_kw=None, tol=1e-5):
    sc = scenes.make_scene(3, B=B, **scene_kw)
    n = sc["xinit"].shape[0]
    for form_dev, form_orc in ((1, 0), (0, 1)):
        d = solver.default_dims(**dims_kw, qp_tol=tol, riccati_form=form_dev)
        s = solver.BatchedSolver(d, B_max=n)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); g = s.get(); ki = s.kernel_info(); s.close()
        pb = O.problem(**(orc_kw or dims_kw), qp_tol=tol, riccati_form=form_orc)
        xt, ut, o = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(n, -1), sc["params"].reshape(n, -1))
        ok = (g["exit_code"] == 1) & (o["exit_code"] == 1)
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        print(name, "tol", tol, "device form", form_dev, "oracle form", form_orc, "exit mism", int((g["exit_code"] != o["exit_code"]).sum()), "sqp mism", int((g["sqp_iter"] != o["sqp_iter"]).sum()),
              "ipm mism", int((g["qp_iter_total"][ok] != o["qp_iter_total"][ok]).sum()), "max rel", float((np.abs(g["xtraj"][ok] - xt[ok]) / sx).max()) if ok.any() else None, "|", ki[:90], flush=True)
for tol in (1e-5, 1e-9):
    run("cfg2", dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8), 64, tol=tol)
    run("cfg1", dict(N=20, S=5, n_lin=0, M=4), dict(N=20, M=4, guidance=False), 1, tol=tol)
    run("cfg4", dict(N=20, S=5, n_lin=12, M=12), dict(N=20, M=12), 64, tol=tol)
    run("cfg5", dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), dict(N=20, M=8, slack=True, n_scenario=24), 32, tol=tol)
    run("cfg3", dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), dict(N=30, M=8, slack=True, n_decomp=12), 64, tol=tol)
    run("cfg3_mpcc", dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), dict(N=30, M=8, slack=True, n_decomp=12), 64, tol=tol)
PY
timeout 900 python tools/tight_flip.py --out gpurun_out/round6_tight_flip.json > gpurun_out/r6_tight_flip.log 2>&1
timeout 600 python -m pytest tests/test_cpp_omp.py -x -q -m gpu -s > gpurun_out/r6_omp.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_suite.log 2>&1
tail -3 gpurun_out/r6_gpu_suite.log
