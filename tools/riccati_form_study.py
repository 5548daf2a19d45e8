"""Numerics study (CPU, oracle only; tools/ script, not product): what does stopping the Riccati elimination after the two input columns --
P_k = F_xx - Lxu Lxu^T used as it is, F = Hh + [B A]^T P [B A] -- instead of re-factorising P_k (square-root form) do to the RTI iterates?
Both forms in the SAME interior-point method (oracle/qp_ipm.c, orc_problem::riccati_form), on bench scenes: exit-code / iteration-count
flips and the relative per-stage trajectory difference.  Decides whether the HIP kernels may drop the five state pivots per stage.
Usage: python tools/riccati_form_study.py [n_scenes] > profiles/round5_riccati_form_study.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_lib as O
from mpc_planner_amd import scenes
cfg, n, qp_tol, form = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[5])
kw = {"cfg2": (dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8)),
      "cfg4": (dict(N=20, M=12, B=64), dict(N=20, S=5, n_lin=12, M=12)),
      "cfg3": (dict(N=30, M=8, B=32, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1))}[cfg]
b = scenes.make_batch(range(2000, 2000 + n), workers=8, **kw[0])
B = b["xinit"].shape[0]
pb = O.problem(qp_tol=qp_tol, riccati_form=form, **kw[1])
xt, ut, info = O.solve_batch(pb, b["xinit"], b["x0"].reshape(B, -1), b["params"].reshape(B, -1), num_threads=8)
np.savez(sys.argv[4], xt=xt, ut=ut, **info)
''' % (ROOT, os.path.join(ROOT, "tests"))

import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = []
for cfg in ("cfg2", "cfg4", "cfg3"):
    for tol in (1e-5, 1e-9):
        res = {}
        for form in ("sqrt", "classical"):
            f = f"/tmp/rfs_{cfg}_{form}.npz"
            subprocess.check_call([sys.executable, "-c", CODE, cfg, str(n), str(tol), f, "1" if form == "classical" else "0"])
            res[form] = dict(np.load(f))
        a, c = res["sqrt"], res["classical"]
        both = (a["exit_code"] == 1) & (c["exit_code"] == 1)
        sx = np.maximum(np.abs(a["xt"][both]).max(axis=2, keepdims=True), 1.0)
        rel = (np.abs(a["xt"][both] - c["xt"][both]) / sx).max(axis=(1, 2))
        same_it = a["qp_iter_total"][both] == c["qp_iter_total"][both]
        out.append({"config": cfg, "qp_tol": tol, "trajectories": int(len(both)), "both_successful": int(both.sum()),
                    "exit_code_flips": int((a["exit_code"] != c["exit_code"]).sum()), "sqp_iter_flips": int((a["sqp_iter"] != c["sqp_iter"]).sum()),
                    "ipm_iter_total_flips": int((~same_it).sum()),
                    "rel_traj_diff": {"median": float(np.median(rel)), "p99": float(np.percentile(rel, 99)), "max": float(rel.max()),
                                      "max_where_iteration_counts_agree": float(rel[same_it].max()) if same_it.any() else None,
                                      "share_above_1e-8": float((rel > 1e-8).mean()), "share_above_1e-6": float((rel > 1e-6).mean())}})
        print(json.dumps(out[-1]), file=sys.stderr, flush=True)
print(json.dumps({"what": "square-root Riccati (re-factorise P_k every stage: 7 pivots) vs the same elimination stopped after the input block (2 pivots, P_k kept as the "
                          "Schur complement), same interior-point method and scenes, CPU oracle FP64", "results": out}, indent=1))
