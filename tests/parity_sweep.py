"""Wide parity sweep (robustness check run by hand on the GPU box, not collected by pytest; lives under tests/ because it uses the oracle): many scenes per shape, HIP path vs CPU oracle.
Prints per shape: trajectories, successes, mismatching exit codes / iteration counts, worst relative trajectory difference."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from mpc_planner_amd import scenes, solver

SHAPES = [
    ("cfg1", dict(N=20, M=4, B=16, guidance=False), dict(N=20, S=5, n_lin=0, M=4), {}),
    ("cfg2", dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), {}),
    # ~10 % of the guidance trajectories carry a point inside an obstacle's disc: LinearizedConstraints::projectToSafety acts (round-4 verdict, next-8)
    ("cfg2-inside", dict(N=20, M=8, B=64, inside_share=0.1), dict(N=20, S=5, n_lin=8, M=8), {}),
    ("cfg4", dict(N=20, M=12, B=31, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12), {}),
    ("N30", dict(N=30, M=8, B=32), dict(N=30, S=5, n_lin=8, M=8), {}),
    ("cfg3", dict(N=30, M=8, B=32, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), {}),
    ("cfg5", dict(N=20, M=8, B=32, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), {}),
    ("rosnav", dict(N=20, M=12, S=8, B=32, slack=True, n_decomp=12), dict(N=20, S=8, n_lin=12, M=12, n_slk=12, slack=1), {}),
    ("jackal-shape", dict(N=30, M=5, S=3, B=32), dict(N=30, S=3, n_lin=5, M=5), {}),
    ("jackal-default (Gaussian rows)", dict(N=30, M=5, S=3, B=32, chance=True), dict(N=30, S=3, n_lin=5, M=5, row_model=1), dict(oracle=dict(N=30, S=3, n_lin=5, M=0, n_gauss=5))),
    ("jackal-default with the CA cost (CM = 3)", dict(N=30, M=5, S=3, B=32, chance=True), dict(N=30, S=3, n_lin=5, M=5, row_model=1, cost_model=1), dict(oracle=dict(N=30, S=3, n_lin=5, M=0, n_gauss=5, cost_model=1))),
    ("cfg3 as named (CA cost)", dict(N=30, M=8, B=32, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), {}),
]
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for name, skw, pkw, opt in SHAPES:
    batch = scenes.make_batch(range(2000, 2000 + n_scenes), **skw)
    B = batch["xinit"].shape[0]
    pb = O.problem(**opt.get("oracle", pkw))
    xt, ut, info = O.solve_batch(pb, batch["xinit"], batch["x0"].reshape(B, -1), batch["params"].reshape(B, -1))
    ok = info["exit_code"] == 1
    s = solver.BatchedSolver(solver.default_dims(**pkw), B_max=B)
    for mode in (0, 1, 2, 3):                      # every kernel variant the shape has (round 6: the latency variants against the same oracle results)
        if mode and not s.set_latency_mode(mode):
            continue
        s.set_latency_mode(mode)
        s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); g = s.get()
        both = ok & (g["exit_code"] == 1)
        sx = np.maximum(np.abs(xt[both]).max(axis=2, keepdims=True), 1.0)
        err = float((np.abs(g["xtraj"][both] - xt[both]) / sx).max()) if both.any() else 0.0
        print(json.dumps(dict(shape=name, latency_mode=mode, trajectories=int(B), success=int(ok.sum()), exit_mismatch=int((g["exit_code"] != info["exit_code"]).sum()),
                              sqp_iter_mismatch=int((g["sqp_iter"] != info["sqp_iter"]).sum()),
                              ipm_iter_mismatch_on_success=int((g["qp_iter_total"][both] != info["qp_iter_total"][both]).sum()),
                              worst_rel_traj_diff=err)), flush=True)
    s.close()
