"""GPU: the C-ABI contracts added in round 5 after the round-4 advisor's findings -- the explicit "copies are not maintained" parameter-sharing mode
(every path that would read the copies fails instead of falling back), tmpc_create_v2 (a caller built against a shorter tmpc_dims), the
compatibility check of tmpc_copy_state, and tmpc_latency_mode_capacity (what the C++ BatchContext uses to pick the tick variant)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(B=64):
    from mpc_planner_amd import scenes
    return scenes.make_scene(3, N=20, M=8, B=B)


def test_strict_param_sharing_fails_instead_of_falling_back():
    from mpc_planner_amd import solver
    sc = _scene()
    dims = solver.default_dims(N=20, S=5, n_lin=8, M=8)
    s = solver.BatchedSolver(dims, B_max=64)
    s.set_batch(sc["xinit"], sc["x0"], sc["params"])
    base = solver.param_sharing_map(sc["params"], dims, 64)
    s.solve(); ref = s.get()
    # the copies really are not maintained: poison the shared columns of every non-base entry
    own = solver.own_parameter_columns(dims)
    p = sc["params"].copy().reshape(64, 20, -1)
    shared = np.setdiff1d(np.arange(p.shape[2]), own)
    p[1:][:, :, shared] = np.nan
    s.set_batch(sc["xinit"], sc["x0"], p.reshape(sc["params"].shape))
    s.set_param_sharing(base, copies_not_maintained=True)
    s.solve(); got = s.get()
    assert np.array_equal(got["exit_code"], ref["exit_code"]) and np.array_equal(got["xtraj"], ref["xtraj"])      # bitwise: the copies were never read
    # a new batch drops the map: the solve must refuse, not read the poisoned copies
    s.set_batch(sc["xinit"], sc["x0"], p.reshape(sc["params"].shape))
    with pytest.raises(solver.TmpcError, match="not in force"):
        s.solve()
    with pytest.raises(solver.TmpcError, match="not in force"):
        s.solve_iterations(1)
    s.set_param_sharing(base, copies_not_maintained=True)
    with pytest.raises(solver.TmpcError, match="lane kernels"):
        s.set_throughput_mode(True)
    with pytest.raises(solver.TmpcError):
        s.debug_profile()
    s.solve(); again = s.get()
    assert np.array_equal(again["xtraj"], ref["xtraj"])
    s.set_param_sharing(None)                                   # cleared: a plain handle again
    s.set_batch(sc["xinit"], sc["x0"], sc["params"]); s.solve()
    assert np.array_equal(s.get()["xtraj"], ref["xtraj"])
    s.close()


def test_create_v2_accepts_the_shorter_struct_of_an_older_header():
    from mpc_planner_amd import solver
    lib = solver.load_library()
    d = solver.default_dims(N=20, S=5, n_lin=8, M=8)
    old_size = solver.TmpcDims.cost_model.offset               # the struct before cost_model / row_model were appended
    # garbage where an old caller's struct ends: must not be read
    raw = (C.c_char * C.sizeof(d)).from_buffer_copy(bytes(d))
    for i in range(old_size, C.sizeof(d)):
        raw[i] = b"\x7f"
    h = C.c_void_p()
    assert lib.tmpc_create_v2(C.byref(h), C.cast(raw, C.POINTER(solver.TmpcDims)), old_size, 4, 0) == 0 and h
    buf = C.create_string_buffer(512)
    lib.tmpc_kernel_info(h, buf, 512)
    assert b"compact" in buf.value or b"fast" in buf.value       # the MPCC + ellipsoid kernels (cost_model 0, row_model 0), not a rejected / Gaussian shape
    lib.tmpc_destroy(h)
    h = C.c_void_p()
    assert lib.tmpc_create_v2(C.byref(h), C.byref(d), old_size - 4, 4, 0) == -1 and not h      # shorter than any revision of the header
    assert lib.tmpc_create_v2(C.byref(h), C.byref(d), old_size + 4, 4, 0) == -1 and not h      # ends inside a field (round-5 advisor): refused, not half-copied
    r5_size = solver.TmpcDims.riccati_form.offset                                               # the round 4-5 header: + cost_model, row_model
    assert lib.tmpc_create_v2(C.byref(h), C.byref(d), r5_size, 4, 0) == 0
    lib.tmpc_destroy(h); h = C.c_void_p()
    assert lib.tmpc_create_v2(C.byref(h), C.byref(d), C.sizeof(d), 4, 0) == 0
    lib.tmpc_destroy(h)


def test_copy_state_refuses_another_stage_model():
    from mpc_planner_amd import solver
    a = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), B_max=4)
    b = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), B_max=4)
    with pytest.raises(solver.TmpcError, match="different shape"):
        b.copy_state_from(a)
    a.close(); b.close()


def test_latency_mode_capacity_orders_the_variants():
    import torch
    from mpc_planner_amd import solver
    s = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8), B_max=8)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    c0, c1, c2 = (s.latency_mode_capacity(m) for m in (0, 1, 2))
    assert c0 == 8 * cus                                         # the compact kernel: eight trajectories per CU
    assert 0 < c2 <= c1 <= c0 and c2 % cus == 0 and c1 % cus == 0
    c3 = s.latency_mode_capacity(3)
    assert c3 == cus                                             # four waves per trajectory: one workgroup per CU is what the variant serves
    s5 = solver.BatchedSolver(solver.default_dims(N=20, S=5, n_lin=8, M=8, row_model=1), B_max=8)   # Gaussian rows: variants 2 and 3 since round 6, no two-wave Riccati variant
    assert s5.latency_mode_capacity(1) == 0 and s5.latency_mode_capacity(2) > 0 and s5.latency_mode_capacity(3) == cus and s5.latency_mode_capacity(0) > 0
    s6 = solver.BatchedSolver(solver.default_dims(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1, cost_model=1), B_max=8)   # curvature-aware cost: the four-wave variant alone
    assert s6.latency_mode_capacity(1) == 0 and s6.latency_mode_capacity(2) == 0 and s6.latency_mode_capacity(3) == cus
    s7 = solver.BatchedSolver(solver.default_dims(N=32, S=5, n_lin=8, M=8), B_max=8)                                    # beyond the parallel-in-time solve's 31 stages: none
    assert s7.latency_mode_capacity(1) == 0 and s7.latency_mode_capacity(2) == 0 and s7.latency_mode_capacity(3) == 0
    s.close(); s5.close(); s6.close(); s7.close()
