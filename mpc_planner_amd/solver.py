"""ctypes binding of the C-ABI (include/tmpc_hip.h, libtmpc_hip.so) + the batched host-side solver object.

`BatchedSolver` is the batch-first counterpart of the reference's `MPCPlanner::Solver`
(mpc_planner_solver/include/mpc_planner_solver/acados_solver_interface.h:93-222): B solver instances'
`_params` (xinit / x0 / all_parameters) in, `_output` (xtraj / utraj) and `_info` out, with the reference's
exit-code convention.  There is NO CPU fallback: if the HIP library or a GPU is missing, construction raises.
"""
import ctypes as C
import os

import numpy as np

NU, NX, NV = 2, 5, 7
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TMPC_HIP_LIBRARY") or os.path.join(_HERE, "libtmpc_hip.so")     # (TMPC_HIP_LIBRARY: lab switch, an A/B build of the same C-ABI)
# The same kernels behind a C-ABI unit built with -DTMPC_LAB_SWITCHES: the only library that reads the TMPC_* kernel-selection overrides from the environment
# (tests that have to reach one kernel family, A/B tools).  The product library (LIB_PATH) ignores the environment.
LAB_LIB_PATH = os.path.join(_HERE, "libtmpc_hip_lab.so")


class TmpcDims(C.Structure):
    _fields_ = [("N", C.c_int32), ("S", C.c_int32), ("n_lin", C.c_int32), ("M", C.c_int32), ("npar", C.c_int32),
                ("n_sqp", C.c_int32), ("qp_iter_max", C.c_int32), ("erk_steps", C.c_int32),
                ("dt", C.c_double), ("qp_tol", C.c_double), ("reg_eps", C.c_double), ("ipm_mu0", C.c_double),
                ("ipm_thr0", C.c_double), ("lb", C.c_double * NV), ("ub", C.c_double * NV),
                ("n_slk", C.c_int32), ("slack", C.c_int32), ("cost_model", C.c_int32), ("row_model", C.c_int32),
                ("riccati_form", C.c_int32)]        # 0: Schur-complement recursion (default), 1: square-root recursion (include/tmpc_hip.h)

    @property
    def nx(self):            # external (model) state / variable counts: the slack model has one more state
        return NX + self.slack

    @property
    def nvar(self):
        return NV + self.slack

    @property
    def nh(self):
        return self.n_lin + self.M + self.n_slk


EXPORTS = ["tmpc_default_dims", "tmpc_default_dims_ex", "tmpc_create", "tmpc_destroy", "tmpc_last_error", "tmpc_set_batch",
           "tmpc_set_batch_device", "tmpc_solve", "tmpc_set_latency_mode", "tmpc_synchronize", "tmpc_get", "tmpc_select_best",
           "tmpc_result_device_ptrs", "tmpc_time_solve", "tmpc_debug_eval_stage", "tmpc_pack_records",
           "tmpc_select_best_records", "tmpc_enable_timing", "tmpc_get_timings", "tmpc_debug_profile",
           "tmpc_linearize_topology", "tmpc_scenario_halfspaces", "tmpc_scenario_support", "tmpc_warmstart", "tmpc_init_with_guidance",
           "tmpc_debug_get_x0", "tmpc_debug_get_params", "tmpc_set_throughput_mode", "tmpc_solve_iterations",
           "tmpc_reset_multipliers", "tmpc_get_stream", "tmpc_kernel_info", "tmpc_set_slots", "tmpc_set_param_sharing", "tmpc_copy_state", "tmpc_scenario_empty_stages", "tmpc_sample_scenarios",
           "tmpc_scenario_discard", "tmpc_scenario_discarded", "tmpc_linearize_topology_ex", "tmpc_clear_slot", "tmpc_gather_best",
           "tmpc_create_v2", "tmpc_set_param_sharing_ex", "tmpc_latency_mode_capacity", "tmpc_has_lane_kernels", "tmpc_debug_lds_passes", "tmpc_debug_poison_lds", "tmpc_has_lab_switches"]

class TmpcError(RuntimeError):
    pass


def has_lane_kernels(lib_path=None):
    """Does this build of the library carry the optional lane-per-trajectory kernels (tmpc_set_throughput_mode)?"""
    lib = load_library(lib_path)
    return hasattr(lib, "tmpc_has_lane_kernels") and lib.tmpc_has_lane_kernels() == 1


_libs = {}


def load_library(path=None):
    """Load libtmpc_hip.so -- or a generated per-configuration library with the same C-ABI (mpc_planner_amd/codegen) --;
    raises (never falls back) if it has not been built."""
    path = os.path.abspath(path) if path else LIB_PATH
    if path in _libs:
        return _libs[path]
    try:
        # If torch is going to be used in this process (device buffers, torch.distributed) its bundled HIP runtime must be the
        # one that gets loaded: libtmpc_hip.so resolves the same libamdhip64 SONAME, and whichever is loaded first serves both.
        # Loading the system runtime first has been seen to leave torch without devices ("No HIP GPUs are available").
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise TmpcError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.tmpc_last_error.restype = C.c_char_p
    lib.tmpc_last_error.argtypes = [C.c_void_p]
    lib.tmpc_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(TmpcDims), C.c_int32, C.c_int32]
    lib.tmpc_destroy.argtypes = [C.c_void_p]
    lib.tmpc_default_dims.argtypes = [C.POINTER(TmpcDims), C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.tmpc_default_dims_ex.argtypes = [C.POINTER(TmpcDims)] + [C.c_int32] * 6
    vp = C.c_void_p
    lib.tmpc_set_batch.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.tmpc_set_batch_device.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.tmpc_solve.argtypes = [vp]
    lib.tmpc_set_latency_mode.argtypes = [vp, C.c_int32]
    lib.tmpc_set_throughput_mode.argtypes = [vp, C.c_int32]
    lib.tmpc_solve_iterations.argtypes = [vp, C.c_int32, C.c_int32]
    lib.tmpc_reset_multipliers.argtypes = [vp]
    lib.tmpc_get_stream.argtypes = [vp, C.POINTER(vp)]
    if hasattr(lib, "tmpc_kernel_info"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_kernel_info.argtypes = [vp, C.c_char_p, C.c_int32]
    if hasattr(lib, "tmpc_set_slots"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_set_slots.argtypes = [vp, vp]
    if hasattr(lib, "tmpc_set_param_sharing"):
        lib.tmpc_set_param_sharing.argtypes = [vp, vp]
    if hasattr(lib, "tmpc_set_param_sharing_ex"):
        lib.tmpc_set_param_sharing_ex.argtypes = [vp, vp, C.c_int32]
        lib.tmpc_latency_mode_capacity.argtypes = [vp, C.c_int32]
        lib.tmpc_create_v2.argtypes = [C.POINTER(C.c_void_p), C.POINTER(TmpcDims), C.c_uint32, C.c_int32, C.c_int32]
    if hasattr(lib, "tmpc_scenario_empty_stages"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_scenario_empty_stages.argtypes = [vp, vp]
    if hasattr(lib, "tmpc_sample_scenarios"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_sample_scenarios.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, vp]
    if hasattr(lib, "tmpc_scenario_discard"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_scenario_discard.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_double]
    if hasattr(lib, "tmpc_linearize_topology_ex"):
        lib.tmpc_linearize_topology_ex.argtypes = [vp, vp, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_double, vp]
    if hasattr(lib, "tmpc_scenario_discarded"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_scenario_discarded.argtypes = [vp, vp]
    if hasattr(lib, "tmpc_copy_state"):        # (absent from reference builds of earlier rounds used in A/B runs)
        lib.tmpc_copy_state.argtypes = [vp, vp]
    if hasattr(lib, "tmpc_gather_best"):
        lib.tmpc_gather_best.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    if hasattr(lib, "tmpc_clear_slot"):
        lib.tmpc_clear_slot.argtypes = [vp, C.c_int32]
    lib.tmpc_synchronize.argtypes = [vp]
    lib.tmpc_get.argtypes = [vp] + [vp] * 8
    lib.tmpc_select_best.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.POINTER(C.c_int32)]
    lib.tmpc_result_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.tmpc_time_solve.argtypes = [vp, C.c_int32, vp]
    lib.tmpc_debug_eval_stage.argtypes = [vp, C.c_int32] + [vp] * 13
    lib.tmpc_pack_records.argtypes = [vp, vp, vp, vp]
    lib.tmpc_select_best_records.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.tmpc_enable_timing.argtypes = [vp, C.c_int32]
    lib.tmpc_get_timings.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.tmpc_debug_profile.argtypes = [vp, vp, C.c_int32]
    lib.tmpc_linearize_topology.argtypes = [vp, vp, vp, vp, C.c_double, vp]
    lib.tmpc_warmstart.argtypes = [vp, vp, vp, vp, C.c_double]
    lib.tmpc_init_with_guidance.argtypes = [vp, vp, vp, vp]
    lib.tmpc_debug_get_x0.argtypes = [vp, vp, vp]
    lib.tmpc_scenario_halfspaces.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_double, C.c_double]
    lib.tmpc_scenario_support.argtypes = [vp, C.c_int32, C.c_double, vp, vp]
    lib.tmpc_debug_get_params.argtypes = [vp, vp]
    _libs[path] = lib
    return lib


def default_dims(N=20, S=5, n_lin=8, M=8, n_slk=0, slack=0, lib_path=None, **opts):
    """lib_path: a generated library (its row / parameter structure overrides n_lin, M, n_slk, slack)."""
    d = TmpcDims()
    load_library(lib_path).tmpc_default_dims_ex(C.byref(d), N, S, n_lin, M, n_slk, int(bool(slack)))
    for k, v in opts.items():
        setattr(d, k, v)
    if d.row_model == 1 and "npar" not in opts:              # Gaussian rows: 6 parameters per obstacle instead of the ellipsoid's 7
        d.npar -= d.M
    return d


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def own_parameter_columns(dims):
    """Indices (within a stage's parameter row) of the entries a planner of a guidance / scenario set sets for itself: the topology
    halfspaces (LinearizedConstraints::setParameters) and the scenario / decomp halfspaces (csrc/tmpc_stage.hpp ip_lin, ip_slk)."""
    base = 8 + dims.slack + 9 * dims.S
    lin = np.arange(base, base + 3 * dims.n_lin)
    disc = base + 3 * dims.n_lin
    slk0 = (disc + 2 + (6 if dims.row_model == 1 else 7) * dims.M) if dims.M > 0 else disc + 1
    return np.concatenate([lin, np.arange(slk0, slk0 + 3 * dims.n_slk)]).astype(int)


def param_sharing_map(params, dims, set_size):
    """base_of for tmpc_set_param_sharing: consecutive groups of `set_size` batch entries (one guidance set each) share the rows of
    their first entry -- CHECKED here: an entry whose shared columns differ from its set's first entry keeps its own rows."""
    B = params.shape[0]
    p = params.reshape(B, dims.N, -1)
    mask = np.ones(p.shape[2], bool); mask[own_parameter_columns(dims)] = False
    out = np.arange(B, dtype=np.int32)
    for s0 in range(0, B, set_size):                                          # set by set: no batch-sized temporaries
        blk = p[s0:s0 + set_size]
        same = (blk[:, :, mask] == blk[0][None, :, mask]).all(axis=(1, 2))
        out[s0:s0 + set_size][same] = s0
    return out


class BatchedSolver:
    """B reference `Solver` instances behind one HIP launch."""

    def __init__(self, dims, B_max, device=0, lib_path=None):
        self.lib = load_library(lib_path)
        self.dims = dims
        self.B_max = int(B_max)
        self.B = 0
        self.device = int(device)
        self._h = C.c_void_p()
        rc = self.lib.tmpc_create(C.byref(self._h), C.byref(dims), self.B_max, int(device))
        if rc != 0:
            raise TmpcError(f"tmpc_create failed with code {rc} (-1 invalid dims, -2 HIP error, -3 no gfx950 device)")
        self.N, self.npar = dims.N, dims.npar

    def close(self):
        if self._h:
            self.lib.tmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise TmpcError(f"{what} failed ({rc}): {self.lib.tmpc_last_error(self._h).decode()}")

    # --- inputs -----------------------------------------------------------------------------------
    def set_batch(self, xinit, x0, params):
        """Host arrays: xinit [B][nx], x0 [B][N+1][nvar], params [B][N][npar] (reference layouts; nx, nvar = 5, 7 or,
        with the slack model, 6, 8)."""
        xinit = np.ascontiguousarray(xinit, np.float64); x0 = np.ascontiguousarray(x0, np.float64)
        params = np.ascontiguousarray(params, np.float64)
        B = xinit.shape[0]
        if xinit.shape != (B, self.dims.nx) or x0.size != B * (self.N + 1) * self.dims.nvar \
                or params.size != B * self.N * self.npar:
            raise ValueError(f"set_batch: expected xinit [{B}][{self.dims.nx}], x0 [{B}][{self.N + 1}][{self.dims.nvar}], "
                             f"params [{B}][{self.N}][{self.npar}]; got {xinit.shape}, {x0.shape}, {params.shape}")
        self._keep = (xinit, x0, params)
        self._check(self.lib.tmpc_set_batch(self._h, B, _p(xinit), _p(x0), _p(params)), "tmpc_set_batch")
        self.B = B

    def set_batch_device(self, B, d_xinit, d_x0, d_params):
        """Raw device pointers (ints), inputs already resident in HBM."""
        self._check(self.lib.tmpc_set_batch_device(self._h, int(B), C.c_void_p(d_xinit), C.c_void_p(d_x0),
                                                   C.c_void_p(d_params)), "tmpc_set_batch_device")
        self.B = int(B)

    # --- solve ------------------------------------------------------------------------------------
    def solve(self, sync=True):
        self._check(self.lib.tmpc_solve(self._h), "tmpc_solve")
        if sync:
            self._check(self.lib.tmpc_synchronize(self._h), "tmpc_synchronize")

    def set_latency_mode(self, on=True):
        """Kernel variant for small control ticks: True / 1 = two waves per trajectory, 2 = the Newton systems solved parallel in time
        (csrc/tmpc_scan.hpp), 3 = four waves per trajectory (round 6: the parallel-in-time solve with its wide phases, the stage evaluation and the
        row passes on 256 lanes; N <= 20), False / 0 = the throughput kernels; returns False if the shape has no such variant."""
        rc = self.lib.tmpc_set_latency_mode(self._h, int(on))
        if rc < 0:
            self._check(rc, "tmpc_set_latency_mode")
        return rc == 0

    KEEP_ITERATE, KEEP_MULTIPLIERS, COMPLETE, NEW_SOLVE = 1, 2, 4, 8

    def solve_iterations(self, n_iter, keep_iterate=False, keep_multipliers=False, complete=True, sync=True, new_solve=False):
        """n_iter RTI iterations per trajectory slot from the state the handle keeps (tmpc_solve_iterations): the reference's
        initializeOneIteration / solveOneIteration / completeOneIteration protocol and multipliers carried across ticks."""
        flags = (self.KEEP_ITERATE if keep_iterate else 0) | (self.KEEP_MULTIPLIERS if keep_multipliers else 0) | (self.COMPLETE if complete else 0) \
            | (self.NEW_SOLVE if new_solve else 0)      # new_solve: first call of a new Solver::solve(): loop exits of the previous solve do not carry over
        self._check(self.lib.tmpc_solve_iterations(self._h, int(n_iter), flags), "tmpc_solve_iterations")
        if sync:
            self.synchronize()

    def set_slots(self, slots):
        """State slot of every entry of the current batch (tmpc_set_slots); None: entry b uses slot b."""
        if slots is None:
            self._check(self.lib.tmpc_set_slots(self._h, None), "tmpc_set_slots")
            return
        a = np.ascontiguousarray(slots, np.int32)
        assert a.size == self.B
        self._check(self.lib.tmpc_set_slots(self._h, a.ctypes.data_as(C.c_void_p)), "tmpc_set_slots")

    def set_param_sharing(self, base_of, copies_not_maintained=False):
        """Hint (tmpc_set_param_sharing): entry b's parameter rows equal entry base_of[b]'s except for its own topology / scenario
        halfspace rows; the kernels read the rest from base_of[b] (same results, 1/64 of a guidance set's parameter traffic).  None clears.
        copies_not_maintained (tmpc_set_param_sharing_ex, TMPC_SHARE_COPIES_NOT_MAINTAINED): the caller writes a set's shared rows into the base
        entry only; every path that would read the other entries' copies then fails instead of falling back."""
        if base_of is None:
            self._check(self.lib.tmpc_set_param_sharing(self._h, None), "tmpc_set_param_sharing")
            return
        a = np.ascontiguousarray(base_of, np.int32)
        assert a.size == self.B
        self._check(self.lib.tmpc_set_param_sharing_ex(self._h, a.ctypes.data_as(C.c_void_p), 1 if copies_not_maintained else 0), "tmpc_set_param_sharing_ex")

    def latency_mode_capacity(self, mode):
        """Trajectories one launch of kernel variant `mode` holds resident on this device (tmpc_latency_mode_capacity); 0: no such variant."""
        rc = self.lib.tmpc_latency_mode_capacity(self._h, int(mode))
        if rc < 0:
            self._check(rc, "tmpc_latency_mode_capacity")
        return rc

    def clear_slot(self, slot):
        """Forget one slot's persistent state (tmpc_clear_slot): the slot's next solve_iterations starts like a fresh capsule."""
        self._check(self.lib.tmpc_clear_slot(self._h, int(slot)), "tmpc_clear_slot")

    def copy_state_from(self, other):
        self._check(self.lib.tmpc_copy_state(self._h, other._h), "tmpc_copy_state")

    def debug_poison_lds(self):
        """Test aid: fill every CU's LDS with NaN bit patterns (tmpc_debug_poison_lds): a kernel that reads a word it never wrote shows."""
        self._check(self.lib.tmpc_debug_poison_lds(self._h), "tmpc_debug_poison_lds")

    def kernel_info(self):
        """Which solve kernel the handle dispatches and how it is launched (text)."""
        buf = C.create_string_buffer(512)
        n = self.lib.tmpc_kernel_info(self._h, buf, 512)
        return buf.value.decode() if n >= 0 else ""

    def stream_ptr(self):
        """hipStream_t of the handle (as an integer), e.g. for torch.cuda.ExternalStream."""
        st = C.c_void_p()
        self._check(self.lib.tmpc_get_stream(self._h, C.byref(st)), "tmpc_get_stream")
        return st.value or 0

    def reset_multipliers(self):
        self._check(self.lib.tmpc_reset_multipliers(self._h), "tmpc_reset_multipliers")

    def set_throughput_mode(self, on=True):
        """Lane-per-trajectory kernels for large batches (allocates the HBM workspace for B_max trajectories on first use)."""
        self._check(self.lib.tmpc_set_throughput_mode(self._h, int(bool(on))), "tmpc_set_throughput_mode")

    def synchronize(self):
        self._check(self.lib.tmpc_synchronize(self._h), "tmpc_synchronize")

    def time_solve(self, reps):
        ms = np.zeros(reps, np.float32)
        self._check(self.lib.tmpc_time_solve(self._h, int(reps), _p(ms)), "tmpc_time_solve")
        return ms

    # --- outputs ----------------------------------------------------------------------------------
    def get(self):
        B, N = self.B, self.N
        out = dict(xtraj=np.zeros((B, N + 1, self.dims.nx)), utraj=np.zeros((B, N, NU)), pobj=np.zeros(B),
                   exit_code=np.zeros(B, np.int32), qp_status=np.zeros(B, np.int32), sqp_iter=np.zeros(B, np.int32),
                   res_eq=np.zeros(B), qp_iter_total=np.zeros(B, np.int32))
        self._check(self.lib.tmpc_get(self._h, _p(out["xtraj"]), _p(out["utraj"]), _p(out["pobj"]), _p(out["exit_code"]),
                                      _p(out["qp_status"]), _p(out["sqp_iter"]), _p(out["res_eq"]),
                                      _p(out["qp_iter_total"])), "tmpc_get")
        return out

    def select_best(self, first=0, count=None, weight=None, disabled=None):
        count = self.B - first if count is None else count
        w = None if weight is None else np.ascontiguousarray(weight, np.float64)
        dis = None if disabled is None else np.ascontiguousarray(disabled, np.uint8)
        best = C.c_int32(-2)
        self._check(self.lib.tmpc_select_best(self._h, int(first), int(count), _p(w), _p(dis), C.byref(best)),
                    "tmpc_select_best")
        return best.value

    def enable_timing(self, max_records):
        self._check(self.lib.tmpc_enable_timing(self._h, int(max_records)), "tmpc_enable_timing")

    def get_timings(self, capacity=4096):
        ms = np.zeros(capacity, np.float32); n = C.c_int32(0)
        self._check(self.lib.tmpc_get_timings(self._h, _p(ms), capacity, C.byref(n)), "tmpc_get_timings")
        return ms[:n.value].copy()

    def pack_records(self, d_records, d_guidance_id=None, d_weight=None):
        """d_*: raw device pointers (ints). Packs {f64 objective, i32 exit_code, i32 guidance_id} per trajectory."""
        self._check(self.lib.tmpc_pack_records(self._h, C.c_void_p(d_records),
                                               C.c_void_p(d_guidance_id) if d_guidance_id else None,
                                               C.c_void_p(d_weight) if d_weight else None), "tmpc_pack_records")

    def select_best_records(self, d_records, n_ranks, n_scenes, per_rank, d_best):
        self._check(self.lib.tmpc_select_best_records(self._h, C.c_void_p(d_records), int(n_ranks), int(n_scenes),
                                                      int(per_rank), C.c_void_p(d_best)), "tmpc_select_best_records")

    def gather_best(self, d_best, n_sets, set_size, d_xtraj, d_utraj, index_offset=0):
        """The winners' trajectories of every set in one compact device buffer (tmpc_gather_best; raw device pointers)."""
        self._check(self.lib.tmpc_gather_best(self._h, C.c_void_p(d_best), int(n_sets), int(set_size), int(index_offset),
                                              C.c_void_p(d_xtraj), C.c_void_p(d_utraj)), "tmpc_gather_best")

    def linearize_topology(self, d_obstacle_pos, d_scene_of, d_state_x, robot_radius, d_is_original=None):
        """Device LinearizedConstraints::update + setParameters (raw device pointers); modifies the batch params in place."""
        self._check(self.lib.tmpc_linearize_topology(self._h, C.c_void_p(d_obstacle_pos), C.c_void_p(d_scene_of),
                                                     C.c_void_p(d_state_x), float(robot_radius),
                                                     C.c_void_p(d_is_original) if d_is_original else None),
                    "tmpc_linearize_topology")

    def linearize_topology_ex(self, d_obstacle_pos, n_obstacles, d_scene_of, d_state_x, robot_radius, d_obstacle_radius=None,
                              d_static_halfspaces=None, n_static=0, d_is_original=None):
        """The whole of LinearizedConstraints::update / setParameters on device (tmpc_linearize_topology_ex): fewer obstacles than rows,
        static halfspace rows (`add_halfspaces`), per-obstacle radii (the `_use_guidance == false` branch)."""
        vp = lambda p_: C.c_void_p(p_) if p_ else None
        self._check(self.lib.tmpc_linearize_topology_ex(self._h, vp(d_obstacle_pos), int(n_obstacles), vp(d_obstacle_radius),
                                                        vp(d_static_halfspaces), int(n_static), C.c_void_p(d_scene_of), C.c_void_p(d_state_x),
                                                        float(robot_radius), vp(d_is_original)), "tmpc_linearize_topology_ex")

    def scenario_halfspaces(self, d_samples, n_pts, n_rows, d_scene_of, d_state_x, radius, disc_offset=0.0):
        """Device scenario -> halfspace reduction of SH-MPC (raw device pointers; samples [n_scenes][N][n_pts][2]);
        modifies the batch params in place."""
        self._check(self.lib.tmpc_scenario_halfspaces(self._h, C.c_void_p(d_samples), int(n_pts), int(n_rows),
                                                      C.c_void_p(d_scene_of), C.c_void_p(d_state_x), float(radius),
                                                      float(disc_offset)), "tmpc_scenario_halfspaces")

    def scenario_support(self, n_scenarios, tol=1e-6):
        """Support of every trajectory's solution (distinct scenarios with an active row; tmpc_scenario_support) after a solve on
        rows built by scenario_halfspaces.  Returns (support [B], active_rows [B]) as numpy int32."""
        import torch
        out = torch.empty((2, self.B), dtype=torch.int32, device=f"cuda:{self.device}")
        self._check(self.lib.tmpc_scenario_support(self._h, int(n_scenarios), float(tol), C.c_void_p(out[0].data_ptr()),
                                                   C.c_void_p(out[1].data_ptr())), "tmpc_scenario_support")
        self.synchronize()
        o = out.cpu().numpy()
        return o[0], o[1]

    def scenario_support_async(self, n_scenarios, tol, d_support, d_active_rows):
        """tmpc_scenario_support on raw device pointers (int32 [B] each), stream-ordered on the handle's stream, no synchronisation."""
        self._check(self.lib.tmpc_scenario_support(self._h, int(n_scenarios), float(tol), C.c_void_p(d_support), C.c_void_p(d_active_rows)),
                    "tmpc_scenario_support")

    def sample_scenarios(self, d_pred, d_prob, n_solvers, n_obstacles, n_modes, n_scenarios, seed, d_samples):
        """Device scenario sampler (tmpc_sample_scenarios): raw device pointers; d_samples [n_solvers][N][n_obstacles * n_scenarios][2]."""
        self._check(self.lib.tmpc_sample_scenarios(self._h, C.c_void_p(d_pred), C.c_void_p(d_prob), int(n_solvers), int(n_obstacles), int(n_modes),
                                                   int(n_scenarios), C.c_uint64(int(seed)), C.c_void_p(d_samples)), "tmpc_sample_scenarios")

    def scenario_discard(self, d_samples, n_pts, n_scenarios, n_discard, d_scene_of, radius):
        """Scenario removal for the current batch (tmpc_scenario_discard); the next scenario_halfspaces leaves the discarded scenarios out."""
        self._check(self.lib.tmpc_scenario_discard(self._h, C.c_void_p(d_samples), int(n_pts), int(n_scenarios), int(n_discard),
                                                   C.c_void_p(d_scene_of), float(radius)), "tmpc_scenario_discard")

    def scenario_discarded(self, n_scenarios):
        import torch
        out = torch.zeros((self.B, n_scenarios), dtype=torch.uint8, device=f"cuda:{self.device}")
        self._check(self.lib.tmpc_scenario_discarded(self._h, C.c_void_p(out.data_ptr())), "tmpc_scenario_discarded")
        self.synchronize()
        return out.cpu().numpy().astype(bool)

    def scenario_empty_stages(self):
        """Per trajectory: the stages whose sampled halfspaces contradicted each other in the last scenario_halfspaces (empty polygon;
        such a stage keeps the closest halfspaces and the trajectory is not eligible)."""
        import torch
        out = torch.zeros(self.B, dtype=torch.int32, device=f"cuda:{self.device}")
        self._check(self.lib.tmpc_scenario_empty_stages(self._h, C.c_void_p(out.data_ptr())), "tmpc_scenario_empty_stages")
        self.synchronize()
        return out.cpu().numpy()

    def warmstart(self, d_state, d_mode=None, d_src=None, deceleration=3.0):
        """Device warm start of the next tick from the solution held by the handle (raw device pointers)."""
        self._check(self.lib.tmpc_warmstart(self._h, C.c_void_p(d_state), C.c_void_p(d_mode) if d_mode else None,
                                            C.c_void_p(d_src) if d_src else None, float(deceleration)), "tmpc_warmstart")

    def init_with_guidance(self, d_gpos, d_gvel, d_enabled=None):
        self._check(self.lib.tmpc_init_with_guidance(self._h, C.c_void_p(d_gpos), C.c_void_p(d_gvel),
                                                     C.c_void_p(d_enabled) if d_enabled else None), "tmpc_init_with_guidance")

    def debug_get_x0(self):
        x0 = np.zeros((self.B, self.N + 1, self.dims.nvar)); xinit = np.zeros((self.B, self.dims.nx))
        self._check(self.lib.tmpc_debug_get_x0(self._h, _p(x0), _p(xinit)), "tmpc_debug_get_x0")
        return x0, xinit

    def debug_get_params(self):
        out = np.zeros((self.B, self.N, self.npar))
        self._check(self.lib.tmpc_debug_get_params(self._h, _p(out)), "tmpc_debug_get_params")
        return out

    def result_device_ptrs(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self.lib.tmpc_result_device_ptrs(self._h, C.byref(a), C.byref(b)), "tmpc_result_device_ptrs")
        return a.value, b.value

    # --- debug ------------------------------------------------------------------------------------
    PHASES = ["linearise", "residuals", "barrier_hessian", "riccati_factor", "rhs", "riccati_solve", "row_passes",
              "update", "final", "total"]

    def debug_profile(self):
        cyc = np.zeros(10, np.int64)
        self._check(self.lib.tmpc_debug_profile(self._h, _p(cyc), 10), "tmpc_debug_profile")
        return dict(zip(self.PHASES, cyc.tolist()))

    def debug_eval_stage(self, z, p, pi=None, lamh=None):
        z = np.ascontiguousarray(z, np.float64).reshape(-1, self.dims.nvar); n = z.shape[0]
        p = np.ascontiguousarray(p, np.float64).reshape(n, self.npar)
        nh = self.dims.nh            # rows in the reference's order [topology | ellipsoids | decomp/scenario rows]
        pi = None if pi is None else np.ascontiguousarray(pi, np.float64).reshape(n, NX)
        lamh = None if lamh is None else np.ascontiguousarray(lamh, np.float64).reshape(n, nh)
        o = dict(cost=np.zeros(n), cost_grad=np.zeros((n, NV)), cost_hess=np.zeros((n, NV, NV)), h=np.zeros((n, nh)),
                 h_jac=np.zeros((n, nh, NV)), x_next=np.zeros((n, NX)), x_jac=np.zeros((n, NX, NV)),
                 lag_hess=np.zeros((n, NV, NV)), mirror=np.zeros((n, NV, NV)))
        self._check(self.lib.tmpc_debug_eval_stage(self._h, n, _p(z), _p(p), _p(pi), _p(lamh), _p(o["cost"]),
                                                   _p(o["cost_grad"]), _p(o["cost_hess"]), _p(o["h"]), _p(o["h_jac"]),
                                                   _p(o["x_next"]), _p(o["x_jac"]), _p(o["lag_hess"]), _p(o["mirror"])),
                    "tmpc_debug_eval_stage")
        return o


def optimize_batch(solver, scene_batch, tmpc_consistency_weight=None):
    """Batched counterpart of GuidanceConstraints::optimize (guidance_constraints.cpp:264-388) for one launch batch:
    load every local planner's parameters + warm start, solve all of them in one launch, then pick the best
    planner per scene (FindBestPlanner :416-434).  Returns (results dict, best index per scene)."""
    solver.set_batch(scene_batch["xinit"], scene_batch["x0"], scene_batch["params"])
    solver.solve()
    res = solver.get()
    scene_of = scene_batch.get("scene_of")
    if scene_of is None:
        return res, np.array([solver.select_best()])
    n_scenes = int(scene_of.max()) + 1
    best = np.zeros(n_scenes, np.int32)
    for s in range(n_scenes):
        idx = np.nonzero(scene_of == s)[0]
        best[s] = solver.select_best(first=int(idx[0]), count=len(idx))
    return res, best


def optimize_scenarios(solver, xinit, x0, params, n_iter=None, scenario=None):
    """Batched counterpart of ScenarioConstraints::optimize (scenario_constraints.cpp:58-108): the P parallel scenario solvers
    (each a copy of the main solver with its own scenario halfspaces in `params` [P][N][npar]; copying the main solver and
    scenario_module.setParameters happen on the caller's side, e.g. modules.halfspace_rows_set_parameters) are solved together,
    driven ONE RTI iteration at a time like the scenario module drives its solver (initializeOneIteration, solveOneIteration x n
    with the loop exit on qp_status != 0, completeOneIteration; :85), then the selection of :93-107: lowest objective among exit
    code 1 (init 1e9, strict '<': lowest index wins ties).  Returns (results dict, best index or -1, exit code the reference
    returns: the best solver's, or the first solver's when none succeeded).

    scenario = dict(d_samples, n_pts, n_rows, d_scene_of, d_state_x, radius, n_scenarios[, disc_offset, tol, max_support, n_discard]) builds
    the rows on device from the sampled scenarios (tmpc_scenario_halfspaces, scenario_module.update + setParameters) and adds the
    support bookkeeping of ScenarioSolver (scenario_constraints.h:38-40): res["support"], res["active_rows"], and with max_support
    (the bound on the support of the solution AFTER the removal, i.e. NOT counting the n_discard removed scenarios: the removed ones
    enter the certificate separately -- the sample size has to come from modules.scenario_sample_size(risk, max_support=max_support,
    removed=n_discard), which is what makes `support <= max_support` certify the risk; tests/test_host_scenario_bound.py)
    res["scenario_status"] (0 = within the bound the sample size was chosen for, 1 = support exceeded: no certificate, 2 = a stage's
    scenario halfspaces contradicted each other -- an empty polygon; res["empty_polygon_stages"] counts them) -- a solver
    with status 1 is then not eligible as the best one."""
    n_iter = solver.dims.n_sqp if n_iter is None else int(n_iter)
    solver.set_batch(xinit, x0, params)                               # *solver = *_solver; setParameters; loadWarmstart
    if scenario is not None:
        if scenario.get("n_discard"):                                  # scenario removal before the polygons; counts into the bound (scenario_risk(removed=...))
            solver.scenario_discard(scenario["d_samples"], scenario["n_pts"], scenario["n_scenarios"], scenario["n_discard"],
                                    scenario["d_scene_of"], scenario["radius"])
        solver.scenario_halfspaces(scenario["d_samples"], scenario["n_pts"], scenario["n_rows"], scenario["d_scene_of"],
                                   scenario["d_state_x"], scenario["radius"], scenario.get("disc_offset", 0.0))
    for it in range(n_iter):                                          # every slot stops by itself once its QP reports a status
        solver.solve_iterations(1, keep_iterate=it > 0, keep_multipliers=True, complete=False)
    solver.solve_iterations(0, keep_iterate=True, keep_multipliers=True, complete=True)
    res = solver.get()
    eligible = res["exit_code"] == 1
    if scenario is not None:
        res["support"], res["active_rows"] = solver.scenario_support(scenario["n_scenarios"], scenario.get("tol", 1e-6))
        res["empty_polygon_stages"] = solver.scenario_empty_stages()
        status = np.zeros(len(res["pobj"]), np.int32)
        if scenario.get("max_support") is not None:
            status[res["support"] > scenario["max_support"]] = 1
        status[res["empty_polygon_stages"] > 0] = 2                  # contradictory scenario halfspaces somewhere on the horizon: never eligible
        if scenario.get("max_support") is not None or (status == 2).any():
            res["scenario_status"] = status
        eligible = eligible & (status == 0)
    best, lowest = -1, 1e9
    for i in range(len(res["pobj"])):
        if eligible[i] and res["pobj"][i] < lowest:
            lowest, best = res["pobj"][i], i
    return res, best, int(res["exit_code"][best if best >= 0 else 0])
