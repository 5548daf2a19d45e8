"""A/B of the two Newton-system factorisations of latency mode 2 against the oracle (and each other): block cyclic reduction (tmpc_scan.hpp)
and the two-front block Cholesky (tmpc_btc.hpp, TMPC_SCAN_TWOFRONT=1).  One process per variant (the variant is chosen at tmpc_create).
Usage: python tools/twofront_ab.py            -> JSON lines"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from mpc_planner_amd import scenes, solver
    s = solver.BatchedSolver(solver.default_dims(N=20), B_max=64)
    assert s.set_latency_mode(2)
    pb = O.problem(N=20, S=5, n_lin=8, M=8)
    out = {"variant": sys.argv[2], "scenes": []}
    for scene in (1, 4, 9, 12, 17, 23):
        sc = scenes.make_scene(scene, N=20, M=8, B=64)
        s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); s.enable_timing(8)
        for _ in range(5):
            s.solve(sync=False)
        ms = float(np.median(s.get_timings())); got = s.get()
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(64, -1), sc["params"].reshape(64, -1))
        ok = (info["exit_code"] == 1) & (got["exit_code"] == 1)
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        out["scenes"].append({"scene": scene, "kernel_ms": ms, "exit_code_mismatch": int((got["exit_code"] != info["exit_code"]).sum()),
                              "sqp_iter_mismatch": int((got["sqp_iter"] != info["sqp_iter"]).sum()),
                              "ipm_iter_mismatch": int((got["qp_iter_total"][ok] != info["qp_iter_total"][ok]).sum()),
                              "max_rel_xtraj": float((np.abs(got["xtraj"][ok] - xt[ok]) / sx).max()) if ok.any() else None, "success": int(ok.sum())})
    print(json.dumps(out), flush=True)
    s.close()
else:
    for name, env in (("cr_two_wave", {}), ("twofront_two_wave", {"TMPC_SCAN_TWOFRONT": "1"}),
                      ("cr_one_wave", {"TMPC_SCAN_WAVES": "1"}), ("twofront_one_wave", {"TMPC_SCAN_TWOFRONT": "1", "TMPC_SCAN_WAVES": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(r.stdout.strip() or json.dumps({"variant": name, "error": r.stderr[-1500:]}), flush=True)
