"""ctypes binding of the CPU oracle (oracle/liboracle_tmpc.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NU, NX, NV = 2, 5, 7
MAX_N, MAX_NH = 32, 40


class Problem(C.Structure):
    _fields_ = [("N", C.c_int), ("S", C.c_int), ("n_lin", C.c_int), ("M", C.c_int), ("npar", C.c_int),
                ("dt", C.c_double), ("n_sqp", C.c_int), ("qp_iter_max", C.c_int), ("qp_tol", C.c_double),
                ("reg_eps", C.c_double), ("ipm_mu0", C.c_double), ("ipm_thr0", C.c_double),
                ("erk_steps", C.c_int), ("lb", C.c_double * NV), ("ub", C.c_double * NV),
                ("n_slk", C.c_int), ("slack", C.c_int), ("lb_slack", C.c_double), ("ub_slack", C.c_double),
                ("n_gauss", C.c_int), ("ipm_tau", C.c_double), ("cost_model", C.c_int),
                ("qp_warm_start", C.c_int), ("ipm_init_box", C.c_int), ("riccati_form", C.c_int), ("model", C.c_int)]

    @property
    def nxe(self):          # model dimensions (array strides): the slack build has one more state
        return NX + self.slack

    @property
    def nve(self):
        return NV + self.slack

    @property
    def nh(self):
        return self.n_lin + self.M + self.n_gauss + self.n_slk


class Info(C.Structure):
    _fields_ = [("pobj", C.c_double), ("res_eq", C.c_double), ("exit_code", C.c_int),
                ("qp_status", C.c_int), ("sqp_iter", C.c_int), ("qp_iter_total", C.c_int)]


class Debug(C.Structure):
    _fields_ = [("W", C.c_double * ((MAX_N + 1) * NV * NV)), ("g", C.c_double * ((MAX_N + 1) * NV)),
                ("BA", C.c_double * (MAX_N * NX * NV)), ("b", C.c_double * (MAX_N * NX)),
                ("h", C.c_double * (MAX_N * MAX_NH)), ("D", C.c_double * (MAX_N * MAX_NH * NV)),
                ("W_raw", C.c_double * ((MAX_N + 1) * NV * NV)), ("z_in", C.c_double * ((MAX_N + 1) * NV)),
                ("pi_in", C.c_double * ((MAX_N + 1) * NX)), ("lamh_in", C.c_double * (MAX_N * MAX_NH)),
                ("dz", C.c_double * ((MAX_N + 1) * NV)), ("pi", C.c_double * ((MAX_N + 1) * NX)),
                ("qp_iters", C.c_int)]


_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib(slack=0):
    """liboracle_tmpc.so (unicycle) or liboracle_tmpc_slack.so (unicycle + slack state): same sources, two builds."""
    slack = int(bool(slack))
    if slack not in _libs:
        path = os.path.join(ORACLE_DIR, "liboracle_tmpc_slack.so" if slack else "liboracle_tmpc.so")
        if not os.path.exists(path):
            build()
        l = C.CDLL(path)
        l.orc_find_best.restype = C.c_int
        assert l.orc_model_nx() == NX + slack
        _libs[slack] = l
    return _libs[slack]


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def problem(N=20, S=5, n_lin=8, M=8, n_slk=0, slack=0, n_gauss=0, **opts):
    pb = Problem()
    lib(slack).orc_problem_init_ex(C.byref(pb), N, S, n_lin, M, n_slk)
    assert pb.slack == int(bool(slack))
    if n_gauss:
        lib(slack).orc_problem_set_gaussian(C.byref(pb), n_gauss)
    goal_stack = opts.pop("goal_stack", None)          # None, or the dynamics model (0 contouring unicycle / 1 SecondOrderUnicycleModel) of the goal-tracking stack
    if goal_stack is not None:
        lib(slack).orc_problem_set_goal_stack(C.byref(pb), int(goal_stack))
    hpipm_like = opts.pop("hpipm_like", None)          # None, or the QP warm-start level (0 / 2) of the HPIPM-like settings
    if hpipm_like is not None:
        lib(slack).orc_problem_set_hpipm_like(C.byref(pb), int(hpipm_like))
    for k, v in opts.items():
        setattr(pb, k, v)
    return pb


def stage_cost(pb, z, p):
    z = np.ascontiguousarray(z, float); p = np.ascontiguousarray(p, float)
    nv = pb.nve
    val = C.c_double(); g = np.zeros(nv); H = np.zeros((nv, nv))
    lib(pb.slack).orc_stage_cost(C.byref(pb), dptr(z), dptr(p), C.byref(val), dptr(g), dptr(H))
    return val.value, g, H


def stage_constraints(pb, z, p):
    z = np.ascontiguousarray(z, float); p = np.ascontiguousarray(p, float)
    nh = pb.nh; nv = pb.nve
    h = np.zeros(nh); J = np.zeros((nh, nv)); H = np.zeros((nh, nv, nv))
    lib(pb.slack).orc_stage_constraints(C.byref(pb), dptr(z), dptr(p), dptr(h), dptr(J), dptr(H))
    return h, J, H


def discrete_dynamics(pb, z):
    z = np.ascontiguousarray(z, float)
    nx, nv = pb.nxe, pb.nve
    xn = np.zeros(nx); J = np.zeros((nx, nv)); H = np.zeros((nx, nv, nv))
    lib(pb.slack).orc_discrete_dynamics(C.byref(pb), dptr(z), dptr(xn), dptr(J), dptr(H))
    return xn, J, H


def continuous_dynamics(z, slack=0):
    z = np.ascontiguousarray(z, float); f = np.zeros(NX + slack)
    lib(slack).orc_continuous_dynamics(dptr(z), dptr(f))
    return f


def mirror(W, eps=1e-4):
    W = np.array(W, float, order="C"); n = W.shape[0]
    lib().orc_mirror(dptr(W), n, C.c_double(eps))
    return W


def solve(pb, xinit, x0, params, debug_iter=None):
    xinit = np.ascontiguousarray(xinit, float); x0 = np.ascontiguousarray(x0, float)
    params = np.ascontiguousarray(params, float)
    assert xinit.size == pb.nxe and x0.size == (pb.N + 1) * pb.nve and params.size == pb.N * pb.npar
    xt = np.zeros((pb.N + 1, pb.nxe)); ut = np.zeros((pb.N, NU)); info = Info()
    if debug_iter is None:
        lib(pb.slack).orc_solve(C.byref(pb), dptr(xinit), dptr(x0), dptr(params), dptr(xt), dptr(ut), C.byref(info))
        return xt, ut, info
    dbg = Debug()
    lib(pb.slack).orc_solve_debug(C.byref(pb), dptr(xinit), dptr(x0), dptr(params), dptr(xt), dptr(ut),
                          C.byref(info), C.byref(dbg), int(debug_iter))
    return xt, ut, info, dbg


def solve_carry(pb, xinit, x0, params, n_iter, pi, lamh):
    """One solve with multipliers carried in and out (pi [(N+1)*NX], lamh [N*MAX_NH] are updated in place)."""
    xinit = np.ascontiguousarray(xinit, float); x0 = np.ascontiguousarray(x0, float); params = np.ascontiguousarray(params, float)
    assert pi.size == (pb.N + 1) * NX and lamh.size == pb.N * MAX_NH and pi.flags.c_contiguous and lamh.flags.c_contiguous
    xt = np.zeros((pb.N + 1, pb.nxe)); ut = np.zeros((pb.N, NU)); info = Info()
    lib(pb.slack).orc_solve_carry(C.byref(pb), dptr(xinit), dptr(x0), dptr(params), int(n_iter), dptr(pi), dptr(lamh),
                                  dptr(xt), dptr(ut), C.byref(info))
    return xt, ut, info


def solve_batch(pb, xinit, x0, params, num_threads=0):
    B = xinit.shape[0]
    xinit = np.ascontiguousarray(xinit, float); x0 = np.ascontiguousarray(x0, float)
    params = np.ascontiguousarray(params, float)
    assert xinit.size == B * pb.nxe and x0.size == B * (pb.N + 1) * pb.nve and params.size == B * pb.N * pb.npar
    xt = np.zeros((B, pb.N + 1, pb.nxe)); ut = np.zeros((B, pb.N, NU)); infos = (Info * B)()
    nt = num_threads or os.cpu_count()
    lib(pb.slack).orc_solve_batch(C.byref(pb), B, dptr(xinit), dptr(x0), dptr(params), dptr(xt), dptr(ut), infos, nt)
    out = dict(pobj=np.array([i.pobj for i in infos]), res_eq=np.array([i.res_eq for i in infos]),
               exit_code=np.array([i.exit_code for i in infos], np.int32),
               qp_status=np.array([i.qp_status for i in infos], np.int32),
               sqp_iter=np.array([i.sqp_iter for i in infos], np.int32),
               qp_iter_total=np.array([i.qp_iter_total for i in infos], np.int32))
    return xt, ut, out


def find_best(objective, exit_code, disabled=None):
    objective = np.ascontiguousarray(objective, float)
    exit_code = np.ascontiguousarray(exit_code, np.int32)
    B = len(objective)
    dis = None if disabled is None else np.ascontiguousarray(disabled, np.uint8).ctypes.data_as(C.POINTER(C.c_ubyte))
    return lib().orc_find_best(B, dptr(objective), exit_code.ctypes.data_as(C.POINTER(C.c_int)), dis)
