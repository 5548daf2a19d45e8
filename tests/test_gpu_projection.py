"""GPU: LinearizedConstraints::projectToSafety where results are compared (round-4 verdict, next-8 / weak #10).  Every other parity scene hands the
solver collision-free guesses, so the projection (linearized_constraints.cpp:130-148; Douglas-Rachford operator of the absent ros_tools, restated:
DESIGN U10) was the identity wherever the device was compared with the oracle.  scenes.make_scene(inside_share=0.1) moves one guidance point of ~10 %
of the trajectories inside an obstacle's disc; here: (i) the device-built rows equal the host mirror's on those scenes, (ii) the geometric property
the survey asks for -- the projected point, recovered from the DEVICE's rows, keeps |p - o| >= r from every obstacle --, (iii) the solve on these
scenes matches the oracle like every other parity scene."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R_DISC = 1e-3 + 0.325


def _batch(n=6):
    from mpc_planner_amd import scenes
    return scenes.make_batch(range(4100, 4100 + n), N=20, M=8, B=64, inside_share=0.1)


def test_device_rows_equal_host_mirror_and_keep_clear_of_every_disc():
    import torch
    from mpc_planner_amd import solver
    b = _batch()
    B = b["xinit"].shape[0]
    dims = solver.default_dims(N=20, S=5, n_lin=8, M=8)
    own = solver.own_parameter_columns(dims)
    start = b["params"].copy().reshape(B, 20, -1)
    start[:, :, own] = -7.0                                                    # the device must rebuild every topology row
    s = solver.BatchedSolver(dims, B_max=B)
    s.set_batch(b["xinit"], b["x0"], start.reshape(b["params"].shape))
    dev = torch.device("cuda")
    t_ob = torch.from_numpy(b["obstacle_pos"]).to(dev); t_sc = torch.from_numpy(b["scene_of"]).to(dev)
    t_sx = torch.from_numpy(np.ascontiguousarray(b["xinit"][::64, 0])).to(dev)
    s.linearize_topology(t_ob.data_ptr(), t_sc.data_ptr(), t_sx.data_ptr(), 0.325, None)
    got = s.debug_get_params().reshape(B, 20, -1)
    want = b["params"].reshape(B, 20, -1)
    np.testing.assert_allclose(got, want, rtol=1e-14, atol=1e-14)
    # geometry from the DEVICE's rows: a_j = (o_j - p) / |o_j - p|, b_j = a_j . o_j - r  =>  p = o_j - d_j a_j; two rows give the distances
    n_checked = 0
    for q in np.flatnonzero(b["inside"]):
        k = int(b["inside_at"][q, 0])
        o = b["obstacle_pos"][q // 64][:, k - 1]
        rows = got[q, k, own].reshape(8, 3)
        np.testing.assert_allclose(rows[:, 0] ** 2 + rows[:, 1] ** 2, 1.0, atol=1e-12)
        np.testing.assert_allclose((rows[:, :2] * o).sum(1) - rows[:, 2], R_DISC, atol=1e-12)       # every halfspace touches its disc
        A2 = np.array([[-rows[0, 0], rows[1, 0]], [-rows[0, 1], rows[1, 1]]])
        if abs(np.linalg.det(A2)) < 1e-6:
            continue
        d01 = np.linalg.solve(A2, o[1] - o[0])
        p = o[0] - d01[0] * rows[0, :2]
        assert (np.hypot(*(p[None] - o).T) >= R_DISC - 1e-9).all(), (q, k)
        assert np.hypot(*(p - b["x0"][q, k, 2:4])) > 1e-3                        # not the identity here
        n_checked += 1
    assert n_checked >= 10
    s.close()


def test_solve_on_scenes_where_the_projection_acts_matches_the_oracle():
    import oracle_lib as O
    from mpc_planner_amd import solver
    from test_gpu_parity import _compare
    b = _batch(4)
    B = b["xinit"].shape[0]
    s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8), B_max=B)
    s.set_batch(b["xinit"], b["x0"], b["params"]); s.solve(); got = s.get(); s.close()
    pb = O.problem(N=20, S=5, n_lin=8, M=8)
    xt, ut, info = O.solve_batch(pb, b["xinit"], b["x0"].reshape(B, -1), b["params"].reshape(B, -1))
    _compare(got, xt, ut, info)
    assert (info["exit_code"][b["inside"]] == 1).mean() > 0.5                    # the perturbed trajectories are solvable problems, not noise
