import os, sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
torch.cuda.init()
from mpc_planner_amd import scenes, solver
small = scenes.make_batch(range(700, 708), N=20, M=8, B=64)
B = small["xinit"].shape[0]
def run(env, **opts):
    for k in ("TMPC_NO_COMPACT",): os.environ.pop(k, None)
    os.environ.update(env)
    s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8, **opts), B_max=B)
    s.set_batch(small["xinit"], small["x0"], small["params"]); s.solve(); r = s.get(); s.close(); return r
for n_sqp, qmax in ((1, 1), (1, 2), (1, 3), (1, 50), (2, 50), (10, 50)):
    a = run({"TMPC_NO_COMPACT": "1"}, n_sqp=n_sqp, qp_iter_max=qmax); b = run({}, n_sqp=n_sqp, qp_iter_max=qmax)
    print(json.dumps(dict(n_sqp=n_sqp, qp_iter_max=qmax, **{k: float(np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float)))) for k in a},
                          nbad=int((a["xtraj"] != b["xtraj"]).any(axis=(1, 2)).sum()))), flush=True)
