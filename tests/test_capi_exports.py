"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol that
include/tmpc_hip.h declares (no compute calls without a GPU), and fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tmpc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tmpc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mpc_planner_amd import solver
    lib = C.CDLL(solver.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"libtmpc_hip.so does not export {n}"
    assert sorted(solver.EXPORTS) == names


def test_default_dims_match_reference_settings():
    from mpc_planner_amd import solver
    d = solver.default_dims(N=20, S=5, n_lin=8, M=8)
    assert d.npar == 135 and d.n_sqp == 10 and d.qp_iter_max == 50 and d.erk_steps == 3
    assert d.dt == 0.2 and d.qp_tol == 1e-5
    assert solver.default_dims(N=20, S=5, n_lin=0, M=4).npar == 83
    assert solver.default_dims(N=20, S=5, n_lin=12, M=12).npar == 175
    # Gaussian rows (tmpc_dims::row_model = 1): 6 parameters per obstacle; mpc_planner_jackal's default map has 82 entries
    # (tests/golden/stage_functions_gaussian.json, made by the reference's own scripts), and the C++ side gets the same from generate_solver
    dj = solver.default_dims(N=30, S=3, n_lin=5, M=5, row_model=1)
    assert dj.npar == 82 and dj.row_model == 1
    from mpc_planner_amd.parameters import define_parameters
    assert define_parameters(3, 5, ellipsoids=False, gaussian=True).length() == 82


def test_generate_solver_writes_the_row_model(tmp_path):
    from mpc_planner_amd import generate_solver
    generate_solver.generate_solver(str(tmp_path), N=30, max_obstacles=5, num_segments=3, gaussian=True)
    hdr = open(tmp_path / "include" / "mpc_planner_solver" / "hip_solver_dims.h").read()
    assert "#define SOLVER_ROW_MODEL 1" in hdr and "#define SOLVER_NP 82" in hdr and "#define SOLVER_M 5" in hdr
    assert "gaussian_obst_4_risk" in open(tmp_path / "config" / "parameter_map.yaml").read()


def test_no_silent_cpu_fallback():
    """Without a GPU the product must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mpc_planner_amd import solver
    with pytest.raises(solver.TmpcError):
        solver.BatchedSolver(solver.default_dims(), B_max=4)


def test_invalid_dims_rejected():
    from mpc_planner_amd import solver
    lib = solver.load_library()
    d = solver.default_dims(); d.npar = 7
    h = C.c_void_p()
    assert lib.tmpc_create(C.byref(h), C.byref(d), 4, 0) == -1 and not h
    for bad in (dict(row_model=2), dict(cost_model=2)):                        # validated before a device is looked for
        d = solver.default_dims(**bad)
        assert lib.tmpc_create(C.byref(h), C.byref(d), 4, 0) == -1 and not h
    d = solver.default_dims(row_model=1, cost_model=1)                         # since round 6 a shape (stage model 3): passes validation; on a CPU box it fails for want of a device
    assert lib.tmpc_create(C.byref(h), C.byref(d), 4, 0) in (0, -3)
    if h:
        lib.tmpc_destroy(h)


def test_create_v2_accepts_only_revision_boundaries():
    """Round-5 advisor: a dims_size that ends inside a field copied part of that field, and a longer struct from a newer header was truncated
    silently.  Sizes are validated before a device is looked for: -1 (TMPC_ERR_INVALID) for anything that is not a revision boundary of
    tmpc_dims, or a longer struct with a non-zero tail; the accepted sizes get as far as the device query (no GPU here: -2 / -3, not -1)."""
    from mpc_planner_amd import solver
    lib = solver.load_library()
    d = solver.default_dims()
    rev = [solver.TmpcDims.cost_model.offset, solver.TmpcDims.riccati_form.offset, C.sizeof(d)]
    h = C.c_void_p()
    for size in rev:
        assert lib.tmpc_create_v2(C.byref(h), C.byref(d), size, 4, 0) != -1 or _has_gpu() is None, size
    for size in (rev[0] - 4, rev[0] + 4, rev[1] + 2, rev[2] - 1, 0, 5000):
        assert lib.tmpc_create_v2(C.byref(h), C.byref(d), size, 4, 0) == -1 and not h, size
    # a newer, longer header: accepted only with a zero tail
    raw = (C.c_char * (C.sizeof(d) + 16)).from_buffer_copy(bytes(d) + b"\0" * 16)
    assert lib.tmpc_create_v2(C.byref(h), C.cast(raw, C.POINTER(solver.TmpcDims)), C.sizeof(d) + 16, 4, 0) != -1 or _has_gpu() is None
    if h:
        lib.tmpc_destroy(h); h = C.c_void_p()
    raw[C.sizeof(d) + 3] = b"\x01"
    assert lib.tmpc_create_v2(C.byref(h), C.cast(raw, C.POINTER(solver.TmpcDims)), C.sizeof(d) + 16, 4, 0) == -1 and not h
    # the Riccati form is validated like the other model fields
    d2 = solver.default_dims(riccati_form=2)
    assert lib.tmpc_create(C.byref(h), C.byref(d2), 4, 0) == -1 and not h


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mpc_planner_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "tmpc_oracle" not in txt, f


def test_param_sharing_map_checks_equality():
    """solver.param_sharing_map (the host side of tmpc_set_param_sharing): sets of consecutive entries share their first entry's rows;
    the planners' own columns (topology / scenario halfspaces) may differ, anything else makes an entry keep its own rows."""
    import numpy as np
    from mpc_planner_amd import solver, scenes
    dims = solver.default_dims(N=20, S=5, n_lin=8, M=8)
    own = solver.own_parameter_columns(dims)
    assert own.tolist() == list(range(8 + 45, 8 + 45 + 24))                   # 8 weights, 9 x 5 spline entries, then 8 x (a1, a2, b)
    b = scenes.make_batch(range(3, 5), N=20, M=8, B=64)
    base = solver.param_sharing_map(b["params"], dims, 64)
    assert base[:64].tolist() == [0] * 64 and base[64:].tolist() == [64] * 64
    p = b["params"].copy().reshape(128, 20, -1)
    p[5, 2, own[3]] += 1.0                                                     # own column: still shared
    p[9, 7, 0] += 1e-9                                                         # a weight: not shared
    base = solver.param_sharing_map(p, dims, 64)
    assert base[5] == 0 and base[9] == 9 and base[10] == 0
    d5 = solver.default_dims(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1)       # scenario rows after the disc offset
    assert solver.own_parameter_columns(d5).tolist() == list(range(8 + 1 + 45 + 1, 8 + 1 + 45 + 1 + 72))


def test_lds_bank_conflict_model_of_the_layout_padding():
    """tmpc_debug_lds_passes: the model tmpc_create ranks the row Jacobians' stage strides with (csrc/tmpc_capi.hip d_load_passes) -- pure host code.
    Pinned against an independent restatement: lane (stage k, sub-lane c) of a wave reads the three entries of its row c + LPS s at k dstride + offset;
    an 8-byte access takes as many passes as the most loaded of the 64 four-byte banks has distinct dwords."""
    import __graft_entry__ as g
    g.build()
    from mpc_planner_amd import solver
    lib = C.CDLL(solver.LIB_PATH)
    lib.tmpc_debug_lds_passes.argtypes = [C.c_int32] * 5
    lib.tmpc_debug_lds_passes.restype = C.c_int

    def model(N, n_pair, nh, threads, dstride):
        lps = (3 if 3 * N <= 64 else 2) if threads == 64 else (6 if N <= 21 else 4)
        nk = min(N, 64 // lps)
        rpl = (nh + 14 + lps - 1) // lps
        total = 0
        for s in range(rpl):
            if lps * s >= nh:
                break
            for part in range(3):
                banks = {}
                for k in range(nk):
                    for c in range(lps):
                        r = c + lps * s
                        a = N * dstride + part
                        if r < nh:
                            off = 2 * r if r < n_pair else 3 * r - n_pair
                            a = k * dstride + off + part if (part < 2 or r >= n_pair) else N * dstride + 2
                        for dw in (2 * a, 2 * a + 1):
                            banks.setdefault(dw % 64, set()).add(dw)
                total += max(len(v) for v in banks.values())
        return total

    # cfg 2's compact layout: 8 packed topology rows + 8 ellipsoid rows = 40 doubles per stage; the padding the library takes is +1
    assert lib.tmpc_debug_lds_passes(20, 8, 16, 64, 40) == 83 and lib.tmpc_debug_lds_passes(20, 8, 16, 64, 41) == 37
    for (N, n_pair, nh, threads) in [(20, 8, 16, 64), (20, 12, 24, 64), (20, 24, 24, 64), (20, 0, 4, 64), (30, 8, 28, 128), (30, 5, 10, 128), (20, 0, 16, 128)]:
        base = 2 * n_pair + 3 * (nh - n_pair)
        for pad in range(6):
            assert lib.tmpc_debug_lds_passes(N, n_pair, nh, threads, base + pad) == model(N, n_pair, nh, threads, base + pad), (N, n_pair, nh, threads, pad)
    assert lib.tmpc_debug_lds_passes(20, 8, 16, 64, 39) < 0 and lib.tmpc_debug_lds_passes(20, 8, 16, 96, 40) < 0      # below the bare stride / no such kernel
