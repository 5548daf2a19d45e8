import sys, json, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from mpc_planner_amd import solver, scenes
import oracle_lib as O
dims = solver.default_dims(N=20)
out = []
for scene in (1, 4, 9):
    sc = scenes.make_scene(scene, N=20, M=8, B=64)
    s = solver.BatchedSolver(dims, B_max=64)
    res = {}
    for mode in (0, 1, 2):
        ok = s.set_latency_mode(mode)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); s.solve()
        s.enable_timing(16)
        for _ in range(10): s.solve(sync=False)
        ms = float(np.median(s.get_timings()))
        res[mode] = (s.get(), ms, ok)
    pb = O.problem(N=20, S=5, n_lin=8, M=8)
    xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(64, -1), sc["params"].reshape(64, -1))
    okm = info["exit_code"] == 1
    rec = {"scene": scene}
    for mode in (0, 1, 2):
        g, ms, okk = res[mode]
        sx = np.maximum(np.abs(xt[okm]).max(axis=2, keepdims=True), 1.0)
        rec[f"mode{mode}"] = {"accepted": bool(okk), "kernel_ms": ms, "exit_mismatch": int((g["exit_code"] != info["exit_code"]).sum()),
                              "sqp_mismatch": int((g["sqp_iter"] != info["sqp_iter"]).sum()),
                              "ipm_iter_mismatch": int((g["qp_iter_total"][okm] != info["qp_iter_total"][okm]).sum()),
                              "ipm_iter_maxdiff": int(np.abs(g["qp_iter_total"][okm] - info["qp_iter_total"][okm]).max()),
                              "max_rel_x": float((np.abs(g["xtraj"][okm] - xt[okm]) / sx).max()), "success": float((g["exit_code"] == 1).mean())}
    print(json.dumps(rec), flush=True)
    s.close()
