"""How far apart are correct implementations of the reference's solver configuration?  (round-3 verdict item 5; CPU only.)
The 10-iteration RTI iterate of the cfg-2 bench scenes, computed by the CPU oracle with four interior-point settings at the reference's
qp_tol = 1e-5 -- tuned (what the kernels run), textbook (0.995, mu0 = thr0 = 1), HPIPM-like cold (mode BALANCE constants, box-interior start),
HPIPM-like with qp_solver_warm_start = 2 (generate_acados_solver.py:173) -- and by the exact-QP active-set RTI (tests/independent_rti.py),
each compared with the tight-tolerance iterate (qp_tol = 1e-9, itself within 2e-7 of the active-set one).  Metric: the north star's
"relative per stage" = max over stages and states of |x - x_ref| / max(1, max_k |x_ref|).
Usage: python tools/iterate_spread.py [n_scenes] > profiles/round4_c_iterate_spread.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import independent_rti as I
import oracle_lib as O
from mpc_planner_amd import scenes

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kw = dict(N=20, S=5, n_lin=8, M=8)
VARIANTS = (("tuned (the kernels' constants: mu0 = thr0 = 0.01, 0.999)", {}),
            ("textbook (mu0 = thr0 = 1, 0.995)", dict(ipm_tau=0.995, ipm_mu0=1.0, ipm_thr0=1.0)),
            ("HPIPM-like, cold QPs (mu0 = 10, thr0 = 0.1, 0.995, box-interior start)", dict(hpipm_like=0)),
            ("HPIPM-like, qp_solver_warm_start = 2 (the reference's setting)", dict(hpipm_like=2)))
rel = {n: [] for n, _ in VARIANTS}
ipm = {n: [0, 0] for n, _ in VARIANTS}
act = []
n_as = 0
for scene in range(70, 70 + n_scenes):
    sc = scenes.make_scene(scene, N=20, M=8, B=64)
    flat = (sc["xinit"], sc["x0"].reshape(64, -1), sc["params"].reshape(64, -1))
    xt9, ut9, info9 = O.solve_batch(O.problem(qp_tol=1e-9, **kw), *flat)
    res = {n: O.solve_batch(O.problem(**kw, **o), *flat) for n, o in VARIANTS}
    full = (info9["exit_code"] == 1) & (info9["sqp_iter"] == 10)
    for n, _ in VARIANTS:
        full &= (res[n][2]["exit_code"] == 1) & (res[n][2]["sqp_iter"] == 10)
    sx = np.maximum(np.abs(xt9).max(axis=2, keepdims=True), 1.0)
    for n, _ in VARIANTS:
        rel[n] += list(((np.abs(res[n][0] - xt9) / sx).max(axis=(1, 2)))[full])
        ipm[n][0] += int(res[n][2]["qp_iter_total"].sum()); ipm[n][1] += int(res[n][2]["sqp_iter"].sum())
    if scene < 72:                                           # the active-set RTI is slow python: 2 x 8 trajectories
        pb = O.problem(**kw)
        for b in np.flatnonzero(full)[:8]:
            xa, ua, _, _ = I.rti_solve(pb, sc["xinit"][b], sc["x0"][b], sc["params"][b])
            act.append(float((np.abs(xa - xt9[b]) / sx[b]).max())); n_as += 1


def stats(v):
    v = np.asarray(v)
    return {"trajectories": int(v.size), "median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "p99": float(np.percentile(v, 99)), "max": float(v.max()),
            "fraction_above_1e-4": float((v > 1e-4).mean()), "fraction_above_1e-3": float((v > 1e-3).mean())}


out = {"what": __doc__.split("Usage")[0].strip(), "scenes": f"cfg 2, scenes 70..{69 + n_scenes} x 64 guidance trajectories (full-length successful solves of every variant only)",
       "reference_iterate": "oracle at qp_tol = 1e-9", "active_set_vs_reference": stats(act) if act else None,
       "variants_at_qp_tol_1e-5": {n: dict(stats(rel[n]), mean_ipm_iterations_per_qp=ipm[n][0] / max(ipm[n][1], 1)) for n, _ in VARIANTS},
       "reading": "At the reference's qp_tol = 1e-5 the 10-iteration iterate depends on HOW the interior-point method reaches that tolerance: implementations "
                  "that are all correct differ by 1e-4 .. 1e-3 relative per stage on a few per cent of the trajectories (and on isolated ones, where ten RTI "
                  "iterations have not settled, by much more).  The north star's 1e-4 against acados is therefore certifiable only at a tight QP tolerance "
                  "(2e-7 at 1e-9); at 1e-5 what is certified is HIP path <-> oracle (same method, ~1e-11) and the distribution above."}
print(json.dumps(out, indent=1))
