"""Kernel time per launch for the BASELINE config shapes (HIP events on the launch stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()            # torch's HIP runtime first (a second in-process HIP initialisation after ours fails)
from mpc_planner_amd import scenes, solver
out = []
for name, kw, dims_kw, per_scene in (
        ("cfg1 MPCC+4 ellipsoids (no guidance), N=20", dict(N=20, M=4, B=64, guidance=False), dict(N=20, S=5, n_lin=0, M=4), 64),
        ("cfg2 T-MPC 8 obstacles, N=20", dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), 64),
        ("cfg4 T-MPC++ 12 obstacles, N=20", dict(N=20, M=12, B=63, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12), 64),
        ("reference default N=30, 8 obstacles (two-wave fast kernel)", dict(N=30, M=8, B=64), dict(N=30, S=5, n_lin=8, M=8), 64),
        ("cfg3 slack model + guidance + ellipsoids + 12 decomp rows, N=30 (two-wave fast kernel)",
         dict(N=30, M=8, B=64, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), 64),
        ("rosnavigation defaults: slack model, 12 topology + 12 decomp rows + 12 ellipsoids, 8 segments, N=20 (runtime-shape two-wave kernel)",
         dict(N=20, M=12, S=8, B=64, slack=True, n_decomp=12), dict(N=20, S=8, n_lin=12, M=12, n_slk=12, slack=1), 64),
        ("jackal defaults: 5 obstacles, 3 segments, N=30 (runtime-shape two-wave kernel)",
         dict(N=30, M=5, S=3, B=64), dict(N=30, S=3, n_lin=5, M=5), 64),
        ("cfg5 SH-MPC 24 scenario halfspaces, slack model, N=20, 32 guidance/scene",
         dict(N=20, M=8, B=32, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), 32)):
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    for n_scenes in ((1, 8, 64) if "cfg5" not in name else (1, 16, 128)):
        batch = scenes.make_batch(range(500, 500 + n_scenes), **kw)
        B = batch["xinit"].shape[0]
        s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=B)
        s.set_batch(batch["xinit"], batch["x0"], batch["params"]); s.solve(); s.solve()
        ms = s.time_solve(5); res = s.get(); s.close()
        out.append(dict(shape=name, B=B, kernel_ms=float(np.median(ms)), solves_per_s=B / (float(np.median(ms)) * 1e-3),
                        success=float((res["exit_code"] == 1).mean()), ipm_per_qp=float(res["qp_iter_total"].sum() / res["sqp_iter"].sum())))
        print(json.dumps(out[-1]), flush=True)

# f-3: scenario -> polygon kernel (8 obstacles x 256 scenarios per stage), 128 scenes x 32 trajectories
if len(sys.argv) == 1 or any("f3" in a for a in sys.argv[1:]):
    kw = dict(N=20, M=8, B=32, slack=True, n_scenario=24)
    scs = [scenes.make_scene(500 + i, **kw) for i in range(16)]
    reps = 8                                                     # 128 scenes: the 16 generated ones repeated
    xinit = np.concatenate([s_["xinit"] for s_ in scs] * reps); x0 = np.concatenate([s_["x0"] for s_ in scs] * reps)
    params = np.concatenate([s_["params"] for s_ in scs] * reps)
    smp = np.stack([np.ascontiguousarray(s_["samples"].transpose(2, 0, 1, 3)).reshape(20, -1, 2) for s_ in scs] * reps)
    B = xinit.shape[0]
    scene_of = np.repeat(np.arange(16 * reps, dtype=np.int32), 32); state_x = np.zeros(16 * reps)
    s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), B_max=B)
    s.set_batch(xinit, x0, params)
    dev = torch.device("cuda")
    t_s = torch.from_numpy(smp).to(dev); t_sc = torch.from_numpy(scene_of).to(dev); t_sx = torch.from_numpy(state_x).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(3):
        s.scenario_halfspaces(t_s.data_ptr(), smp.shape[2], 24, t_sc.data_ptr(), t_sx.data_ptr(), 0.725); s.synchronize()
    import time
    t0 = time.perf_counter()
    for it in range(20):
        s.scenario_halfspaces(t_s.data_ptr(), smp.shape[2], 24, t_sc.data_ptr(), t_sx.data_ptr(), 0.725)
    s.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    pts = B * 19 * smp.shape[2]                                   # sampled positions turned into halfspaces per launch
    print(json.dumps(dict(shape="f-3 scenario->polygon", B=B, kernel_ms=ms, samples_per_launch=pts,
                          algorithmic_GBps=pts * 16 / (ms * 1e-3) / 1e9, unique_sample_bytes=int(smp.nbytes))), flush=True)
    s.close()
