"""Launch the auxiliary device kernels (f-1 topology linearisation, f-2 warm start / guidance init, f-3 scenario reduction,
records + selection) at the bench batch size; run under `rocprofv3 --kernel-trace --stats` to get their durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
from mpc_planner_amd import scenes, solver
dev = torch.device("cuda")
batch = scenes.make_batch(range(64), N=20, M=8, B=64)
B = batch["xinit"].shape[0]
s = solver.BatchedSolver(solver.default_dims(), B_max=B)
s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve()
obst = np.stack([scenes.make_scene(i, N=20, M=8, B=1)["obstacles"]["pos"] for i in range(64)])
t = dict(ob=torch.from_numpy(obst).to(dev), sc=torch.from_numpy(batch["scene_of"]).to(dev), sx=torch.zeros(64, dtype=torch.float64, device=dev),
         st=torch.from_numpy(batch["xinit"]).to(dev), gp=torch.zeros((B, 21, 2), dtype=torch.float64, device=dev),
         gv=torch.ones((B, 21, 2), dtype=torch.float64, device=dev), rec=torch.zeros((B, 2), dtype=torch.int64, device=dev),
         best=torch.zeros(64, dtype=torch.int32, device=dev))
for _ in range(10):
    s.warmstart(t["st"].data_ptr())
    s.init_with_guidance(t["gp"].data_ptr(), t["gv"].data_ptr())
    s.linearize_topology(t["ob"].data_ptr(), t["sc"].data_ptr(), t["sx"].data_ptr(), 0.325)
    s.pack_records(t["rec"].data_ptr(), None)
    s.select_best_records(t["rec"].data_ptr(), 1, 64, 64, t["best"].data_ptr())
s.synchronize(); s.close()
print("done")
