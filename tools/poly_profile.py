"""Segment clocks of one unit of tmpc_scenario_halfspaces (a library built with -DTMPC_POLY_PROFILE prints them): cfg 5's 32 solvers x 20 stages x 2048 samples."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpc_planner_amd import scenes, solver
import torch
sc = scenes.make_scene(500, N=20, M=8, B=32, slack=True, n_scenario=24)
s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), B_max=32)
s.set_batch(sc["xinit"], sc["x0"], sc["params"])
smp = np.ascontiguousarray(np.ascontiguousarray(sc["samples"].transpose(2, 0, 1, 3)).reshape(1, 20, -1, 2))
dev = torch.device("cuda")
t_s = torch.from_numpy(smp).to(dev); t_sc = torch.zeros(32, dtype=torch.int32, device=dev); t_sx = torch.zeros(1, dtype=torch.float64, device=dev)
print("samples per stage", smp.shape[2])
for it in range(2):
    s.scenario_halfspaces(t_s.data_ptr(), smp.shape[2], 24, t_sc.data_ptr(), t_sx.data_ptr(), 0.725); s.synchronize()
s.close()
