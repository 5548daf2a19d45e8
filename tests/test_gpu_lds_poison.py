"""No solve kernel may read LDS it has not written (round 6).  LDS keeps its contents from kernel to kernel; in a fresh test process it mostly
holds zeros, so a kernel that multiplies an unwritten word by zero passes every parity test -- and fails in a process whose earlier kernels left
NaN bit patterns there (bench.py found exactly that in the first four-wave factorisation: 17 % of a launch failed).  tmpc_debug_poison_lds fills
every CU's LDS with signalling NaNs; a solve after it must give the results of the solve before it, bit for bit -- every kernel family."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "cfg2 tuned (8, 8)": (dict(N=20, S=5, n_lin=8, M=8), dict(N=20, M=8), 64),
    "cfg4 (12, 12)": (dict(N=20, S=5, n_lin=12, M=12), dict(N=20, M=12), 64),
    "cfg5 scenario rows, slack": (dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), dict(N=20, M=8, slack=True, n_scenario=24), 32),
    "cfg1": (dict(N=20, S=5, n_lin=0, M=4), dict(N=20, M=4, guidance=False), 1),
    "N = 11": (dict(N=11, S=5, n_lin=8, M=8), dict(N=11, M=8), 16),
    "cfg3 N = 30": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), dict(N=30, M=8, slack=True, n_decomp=12), 32),
    "cfg3 N = 30, CA cost": (dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), dict(N=30, M=8, slack=True, n_decomp=12), 32),
    "jackal default with the CA cost (CM = 3: generic + four-wave kernels)": (dict(N=30, S=3, n_lin=5, M=5, row_model=1, cost_model=1), dict(N=30, M=5, S=3, chance=True), 16),
    "jackal default, Gaussian rows": (dict(N=30, S=3, n_lin=5, M=5, row_model=1), dict(N=30, M=5, S=3, chance=True), 32),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_results_do_not_depend_on_what_the_lds_held(name):
    from mpc_planner_amd import scenes, solver
    dims_kw, scene_kw, B = CASES[name]
    sc = scenes.make_scene(4, B=B, **scene_kw)
    n = sc["xinit"].shape[0]
    modes = [0, 1, 2, 3]
    forms = [0, 1] if not dims_kw.get("row_model") else [0]
    ran = 0
    for form in forms:
        s = solver.BatchedSolver(solver.default_dims(**dims_kw, riccati_form=form), B_max=max(n, 2048))
        for mode in modes:
            if mode and not s.set_latency_mode(mode):
                continue
            s.set_latency_mode(mode)
            s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve(); clean = s.get()
            s.debug_poison_lds()
            s.solve(); dirty = s.get()
            for k in ("exit_code", "qp_status", "sqp_iter", "qp_iter_total", "xtraj", "utraj", "pobj"):
                assert np.array_equal(clean[k], dirty[k]), (name, form, mode, k)
            ran += 1
        if form == 0 and n < 2048:                                      # the compact (persistent) kernels serve launches beyond the fast kernels' resident set
            rep = -(-1536 // n)
            big = [np.tile(sc[k], (rep,) + (1,) * (sc[k].ndim - 1)) for k in ("xinit", "x0", "params")]
            s.set_latency_mode(0)
            s.set_batch(*big); s.solve(); clean = s.get()
            s.debug_poison_lds()
            s.solve(); dirty = s.get()
            for k in ("exit_code", "qp_iter_total", "xtraj"):
                assert np.array_equal(clean[k], dirty[k]), (name, "large launch", k)
            ran += 1
        s.close()
    assert ran >= 2
