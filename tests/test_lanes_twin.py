"""Lane-per-trajectory throughput kernel, CPU side: the kernel's scalar per-lane program (mpc_planner_amd/csrc/tmpc_lanes.hpp),
compiled for the host by tests/cpu_twin (test infrastructure), against the oracle on every BASELINE shape -- exit codes, RTI and
interior-point iteration counts exact, trajectories to rounding.  The device build of the same source is compared on the GPU in
tests/test_gpu_parity.py::test_throughput_mode_matches_oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from mpc_planner_amd import scenes, solver

pytestmark = pytest.mark.lanes          # the optional lane-per-trajectory family: collected only with TMPC_BUILD_LANES=1 (tests/conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(ROOT, "tests", "cpu_twin", "liblanes_twin.so")


def _twin():
    if not os.path.exists(TWIN):
        import __graft_entry__ as g
        g.build_cpu_twin()
    lib = C.CDLL(TWIN)
    lib.lanes_twin_solve.argtypes = [C.POINTER(solver.TmpcDims), C.c_int32] + [C.c_void_p] * 11
    return lib


def twin_solve(dims, xinit, x0, params):
    lib = _twin()
    B, N = xinit.shape[0], dims.N
    xinit = np.ascontiguousarray(xinit, float); x0 = np.ascontiguousarray(x0.reshape(B, -1), float)
    params = np.ascontiguousarray(params.reshape(B, -1), float)
    out = dict(xtraj=np.zeros((B, N + 1, dims.nx)), utraj=np.zeros((B, N, 2)), pobj=np.zeros(B), exit_code=np.zeros(B, np.int32),
               qp_status=np.zeros(B, np.int32), sqp_iter=np.zeros(B, np.int32), res_eq=np.zeros(B), qp_iter_total=np.zeros(B, np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.lanes_twin_solve(C.byref(dims), B, p(xinit), p(x0), p(params), p(out["xtraj"]), p(out["utraj"]), p(out["pobj"]),
                              p(out["exit_code"]), p(out["qp_status"]), p(out["sqp_iter"]), p(out["res_eq"]), p(out["qp_iter_total"]))
    assert rc == 0
    return out


CASES = {
    "cfg1": (dict(N=20, M=4, B=4, guidance=False), dict(N=20, S=5, n_lin=0, M=4)),
    "cfg2": (dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8)),
    "cfg3": (dict(N=30, M=8, B=32, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1)),
    "cfg4": (dict(N=20, M=12, B=32, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12)),
    "cfg5": (dict(N=20, M=8, B=32, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1)),
    "short_horizon": (dict(N=5, M=4, B=8), dict(N=5, S=5, n_lin=4, M=4)),
    # edge shapes: minimal horizon, the oracle's maximal horizon, one row class only, many rows, other segment count
    "N2": (dict(N=2, M=4, B=8), dict(N=2, S=5, n_lin=4, M=4)),
    "N32": (dict(N=32, M=6, B=8), dict(N=32, S=5, n_lin=6, M=6)),
    "ellipsoids_only": (dict(N=20, M=12, B=6, guidance=False), dict(N=20, S=5, n_lin=0, M=12)),
    "many_rows": (dict(N=20, M=12, S=8, B=8, slack=True, n_decomp=12), dict(N=20, S=8, n_lin=12, M=12, n_slk=12, slack=1)),
    "three_segments": (dict(N=30, M=5, S=3, B=8), dict(N=30, S=3, n_lin=5, M=5)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_lane_program_matches_oracle(name):
    skw, pkw = CASES[name]
    dims = solver.default_dims(**pkw)
    pb = O.problem(**pkw)
    n_ok = 0
    for scene in (40, 41, 42):
        sc = scenes.make_scene(scene, **skw)
        B = sc["xinit"].shape[0]
        if name == "cfg2" and scene == 41:          # a few infeasible trajectories: the failure exits must agree too
            pm = sc["pm"]
            for b in (5, 17):
                for j, sg in ((0, 1.0), (1, -1.0)):
                    sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_a1")] = sg
                    sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_a2")] = 0.0
                    sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_b")] = sg * sc["x0"][b, 1:-1, 2] - 5.0
        g = twin_solve(dims, sc["xinit"], sc["x0"], sc["params"])
        xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(B, -1), sc["params"].reshape(B, -1))
        assert (g["exit_code"] == info["exit_code"]).all()
        assert (g["sqp_iter"] == info["sqp_iter"]).all()
        ok = info["exit_code"] == 1
        assert (g["qp_iter_total"][ok] == info["qp_iter_total"][ok]).all() and (g["qp_status"][ok] == info["qp_status"][ok]).all()
        sx = np.maximum(np.abs(xt[ok]).max(axis=2, keepdims=True), 1.0)
        assert (np.abs(g["xtraj"][ok] - xt[ok]) / sx).max() < 1e-8
        assert (np.abs(g["pobj"][ok] - info["pobj"][ok]) / np.maximum(np.abs(info["pobj"][ok]), 1.0)).max() < 1e-8
        assert np.abs(g["res_eq"][ok] - info["res_eq"][ok]).max() < 1e-8
        n_ok += int(ok.sum())
    assert n_ok >= 3
