import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from mpc_planner_amd import scenes, solver
def run(name, dims_kw, scene_kw, B, scene, sl=None, share=0, reps=100):
    sc = scenes.make_scene(scene, B=B, **scene_kw)
    xi, x0, pr = sc["xinit"], sc["x0"], sc["params"]
    if sl: xi, x0, pr = xi[sl], x0[sl], pr[sl]
    n = xi.shape[0]
    ref = None
    for mode in (2, 3):
        s = solver.BatchedSolver(solver.default_dims(**dims_kw), B_max=n)
        s.set_latency_mode(mode); s.set_batch(xi, x0, pr)
        if share: s.set_param_sharing(solver.param_sharing_map(pr, s.dims, share))
        s.solve(); first = s.get()
        bad_launches = 0; worst = 0
        for r in range(reps):
            for _ in range(3): s.solve(sync=False)
            s.solve(); g = s.get()
            d = int((g["exit_code"] != first["exit_code"]).sum()) + int((g["qp_iter_total"] != first["qp_iter_total"]).sum())
            if d or not np.array_equal(g["xtraj"], first["xtraj"]):
                bad_launches += 1; worst = max(worst, d)
        s.close()
        print(name, "mode", mode, "launch groups differing from the first:", bad_launches, "of", reps, "worst integer diffs", worst, flush=True)
        if ref is None: ref = first
        else: print(name, "mode 3 vs 2 integer mismatches", int((first["exit_code"] != ref["exit_code"]).sum()), int((first["qp_iter_total"] != ref["qp_iter_total"]).sum()))
run("cfg4 shared", dict(N=20, S=5, n_lin=12, M=12), dict(N=20, M=12), 4096, 7, slice(0, 512), share=512)
run("cfg5 shared", dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), dict(N=20, M=8, slack=True, n_scenario=24), 32, 7, share=32)
run("cfg2 shared", dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8), 64, 7, share=64)
run("cfg2 512", dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8), 512, 7, share=64)
