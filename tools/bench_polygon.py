"""Timing of the f-3 polygon kernel (tmpc_scenario_halfspaces) against the number of samples per stage.
    python tools/bench_polygon.py [n_scenes_x16]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpc_planner_amd import scenes, solver
import torch

kw = dict(N=20, M=8, B=32, slack=True, n_scenario=24)
scs = [scenes.make_scene(500 + i, **kw) for i in range(16)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
xinit = np.concatenate([s_["xinit"] for s_ in scs] * reps); x0 = np.concatenate([s_["x0"] for s_ in scs] * reps)
params = np.concatenate([s_["params"] for s_ in scs] * reps)
full = np.stack([np.ascontiguousarray(s_["samples"].transpose(2, 0, 1, 3)).reshape(20, -1, 2) for s_ in scs] * reps)
B = xinit.shape[0]
scene_of = np.repeat(np.arange(16 * reps, dtype=np.int32), 32); state_x = np.zeros(16 * reps)
s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), B_max=B)
s.set_batch(xinit, x0, params)
dev = torch.device("cuda")
t_sc = torch.from_numpy(scene_of).to(dev); t_sx = torch.from_numpy(state_x).to(dev)
rng = np.random.default_rng(0)
for n in (256, 512, 1024, 2048):
    pick = np.sort(rng.choice(full.shape[2], n, replace=False))
    smp = np.ascontiguousarray(full[:, :, pick])
    t_s = torch.from_numpy(smp).to(dev)
    for it in range(2):
        s.scenario_halfspaces(t_s.data_ptr(), n, 24, t_sc.data_ptr(), t_sx.data_ptr(), 0.725); s.synchronize()
    t0 = time.perf_counter()
    for it in range(10):
        s.scenario_halfspaces(t_s.data_ptr(), n, 24, t_sc.data_ptr(), t_sx.data_ptr(), 0.725)
    s.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps(dict(n_pts=n, B=B, stages=B * 19, kernel_ms=ms, ns_per_sample=ms * 1e6 / (B * 19 * n))), flush=True)
s.close()
