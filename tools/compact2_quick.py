"""Quick guard before tools/compact2_ab.py: one small launch of the two-wave compact kernel against the fast kernel (bitwise)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_planner_amd import scenes
b = scenes.make_batch(range(800, 802), N=30, M=8, B=64)
import torch
torch.cuda.init()
from mpc_planner_amd import solver
if not os.environ.get('TMPC_HIP_LIBRARY'):
    solver.LIB_PATH = solver.LAB_LIB_PATH      # (round 6: the TMPC_* kernel-selection switches exist in the lab build of the library only)
res = []
for env in (dict(TMPC_NO_COMPACT="1"), dict(TMPC_COMPACT2_MIN_B="0")):
    for k in ("TMPC_NO_COMPACT", "TMPC_COMPACT2_MIN_B"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8), B_max=128)
    print(s.kernel_info(), flush=True)
    s.set_batch(b["xinit"], b["x0"], b["params"]); s.solve(); res.append(s.get()); s.close()
same = {k: bool(np.array_equal(res[0][k], res[1][k], equal_nan=True)) for k in res[0]}
print("bitwise", all(same.values()), same, "success", float((res[1]["exit_code"] == 1).mean()))
sys.exit(0 if all(same.values()) else 1)
