"""TEST INFRASTRUCTURE: a second, structurally different realisation of Solver::solve()'s RTI loop, used to pin the oracle's
10-iteration iterate to something that shares neither its QP solver nor its regularisation code (VERDICT r1, weak point 1).

    oracle/sqp_rti.c + qp_ipm.c          this file
    -----------------------------------  ------------------------------------------------------------------
    Mehrotra interior point              Goldfarb-Idnani DUAL ACTIVE-SET method (exact vertex solution, no barrier,
    stage-wise square-root Riccati       no tolerances tuned on any scene), on the fully CONDENSED dense QP in the
    on the sparse OCP-structured QP      2N inputs (states eliminated through the linearised dynamics)
    MIRROR by cyclic Jacobi (mirror.c)   MIRROR by LAPACK (numpy.linalg.eigh)
    multipliers from the IPM             row multipliers from the active set, dynamics multipliers by the adjoint recursion

Shared on purpose: the stage functions (cost / dynamics / rows with first and second derivatives), which are pinned to the
reference's own python modules by tests/golden/stage_functions*.json, and the RTI protocol of SURVEY Appendix B (full step,
multipliers of the QP, exact Lagrangian Hessian).  After MIRROR every QP is strictly convex, so both QP solvers must return
the same step up to the interior-point tolerance (qp_tol = 1e-5, generate_acados_solver.py:162).
"""
import numpy as np

import oracle_lib as O

NU, NX, NV = 2, 5, 7
INF = 1e10


def mirror(W, eps):
    e, V = np.linalg.eigh(0.5 * (W + W.T))
    e = np.where(np.abs(e) <= eps, eps, np.abs(e))
    return (V * e) @ V.T


def goldfarb_idnani(H, f, C, d, tol=1e-11, max_iter=2000):
    """min 1/2 x'Hx + f'x  s.t.  C x >= d,  H positive definite.  Returns (x, lam >= 0 per row, active set)."""
    n = len(f)
    Hinv = np.linalg.inv(H)
    x = -Hinv @ f
    act, u = [], np.zeros(0)
    for _ in range(max_iter):
        s = C @ x - d
        s[act] = 0.0
        p = int(np.argmin(s))
        if s[p] >= -tol:
            lam = np.zeros(len(d)); lam[act] = u
            return x, lam, act
        npl = C[p]
        u = np.append(u, 0.0)
        while True:                                            # partial steps until constraint p is satisfied
            if act:
                Na = C[act].T                                  # n x q
                M = np.linalg.inv(Na.T @ Hinv @ Na)
                Nstar = M @ Na.T @ Hinv
                z = Hinv @ npl - Hinv @ Na @ (Nstar @ npl)
                r = Nstar @ npl
            else:
                z = Hinv @ npl; r = np.zeros(0)
            t1, drop = np.inf, -1
            for j in range(len(act)):
                if r[j] > 1e-14 and u[j] / r[j] < t1:
                    t1, drop = u[j] / r[j], j
            zn = z @ npl
            t2 = -(C[p] @ x - d[p]) / zn if zn > 1e-14 else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                raise RuntimeError("QP infeasible")
            u[:len(act)] -= t * r
            u[-1] += t
            if np.isfinite(t2):
                x = x + t * z
            if t == t2:
                act.append(p)
                break
            act.pop(drop)
            u = np.delete(u, drop)
    raise RuntimeError("active-set iteration limit")


def rti_solve(pb, xinit, x0, params, n_sqp=None):
    """n_sqp RTI iterations from the warm start x0 [N+1][nv] (fresh multipliers).  Returns xtraj [N+1][nx], utraj [N][2], pobj,
    list of active-set sizes.  The slack model (one more state with slack' = 0, solver_model.py:274-298) is carried as a real
    state: condensing eliminates it through its own dynamics row like every other state, which checks the oracle's / kernels'
    presolve of that state (DESIGN U9) instead of repeating it."""
    NX, NV = pb.nxe, pb.nve
    N, nh = pb.N, pb.nh
    n_sqp = pb.n_sqp if n_sqp is None else n_sqp
    z = np.array(x0, float).reshape(N + 1, NV).copy()
    z[N, :NU] = 0.0
    pi = np.zeros((N + 1, NX)); lamh = np.zeros((N, nh))
    params = np.asarray(params, float).reshape(N, -1)
    # guidance rows <= 0, ellipsoid rows >= 1, decomp / scenario rows <= 0 (module order, as orc_stage_constraints reports them)
    lh = np.concatenate([np.full(pb.n_lin, -np.inf), np.ones(pb.M), np.full(pb.n_slk, -np.inf)])
    uh = np.concatenate([np.zeros(pb.n_lin), np.full(pb.M, np.inf), np.zeros(pb.n_slk)])
    lb, ub = np.array(list(pb.lb) + [pb.lb_slack] * pb.slack), np.array(list(pb.ub) + [pb.ub_slack] * pb.slack)
    sizes = []
    for _ in range(n_sqp):
        W = np.zeros((N + 1, NV, NV)); g = np.zeros((N + 1, NV)); A = np.zeros((N, NX, NV)); b = np.zeros((N, NX))
        rows = []                                                  # (k, coefficient on v_k (7), rhs) meaning  coef . v_k >= rhs
        owner = []                                                 # (k, general row index or -1, +1 lower / -1 upper)
        for k in range(N):
            l, gl, Hl = O.stage_cost(pb, z[k], params[k])
            xn, J, Hd = O.discrete_dynamics(pb, z[k])
            h, D, Hh = O.stage_constraints(pb, z[k], params[k])
            Wk = pb.dt * Hl + np.tensordot(pi[k + 1], Hd, 1) + np.tensordot(lamh[k], Hh, 1)
            W[k] = mirror(Wk, pb.reg_eps); g[k] = pb.dt * gl
            A[k] = J; b[k] = xn - z[k + 1, NU:]
            for r in range(nh):
                if lh[r] > -INF:
                    rows.append((k, D[r], lh[r] - h[r])); owner.append((k, r, 1.0))
                if uh[r] < INF:
                    rows.append((k, -D[r], -(uh[r] - h[r]))); owner.append((k, r, -1.0))
            for j in range(NU if k == 0 else NV):                  # x_0 is fixed, not boxed (DESIGN U2)
                e = np.zeros(NV); e[j] = 1.0
                rows.append((k, e, lb[j] - z[k, j])); owner.append((k, -1, 1.0))
                rows.append((k, -e, -(ub[j] - z[k, j]))); owner.append((k, -1, -1.0))
        W[N, NU:, NU:] = mirror(np.zeros((NX, NX)), pb.reg_eps)
        # ---- condensing: v_k = S_k u + s_k,  u = (du_0 .. du_{N-1}) ----
        nu_tot = NU * N
        S = np.zeros((N + 1, NV, nu_tot)); s = np.zeros((N + 1, NV))
        Gx = np.zeros((NX, nu_tot)); gx = np.asarray(xinit, float)[:NX] - z[0, NU:]
        for k in range(N + 1):
            S[k, NU:] = Gx; s[k, NU:] = gx
            if k < N:
                S[k, :NU, NU * k:NU * k + NU] = np.eye(NU)
                Gx = A[k] @ S[k]; gx = A[k] @ s[k] + b[k]
        H = sum(S[k].T @ W[k] @ S[k] for k in range(N + 1))
        f = sum(S[k].T @ (W[k] @ s[k] + g[k]) for k in range(N + 1))
        C = np.array([c @ S[k] for k, c, _ in rows]); dd = np.array([rhs - c @ s[k] for k, c, rhs in rows])
        free = np.abs(C).max(axis=1) > 0.0                         # rows on a state the inputs cannot move (the pinned slack): constants
        assert (dd[~free] <= 1e-9).all(), "constant row violated"
        u, lam_f, act = goldfarb_idnani(0.5 * (H + H.T), f, C[free], dd[free])
        lam = np.zeros(len(dd)); lam[free] = lam_f
        sizes.append(len(act))
        v = np.array([S[k] @ u + s[k] for k in range(N + 1)])
        # ---- multipliers: rows from the active set, dynamics by the adjoint recursion ----
        rowgrad = np.zeros((N + 1, NV)); lam_gen = np.zeros((N, nh))
        for (k, c, _), (_, r, sg), lm in zip(rows, owner, lam):
            rowgrad[k] += lm * c
            if r >= 0:
                lam_gen[k, r] += -sg * lm                          # (lam_upper - lam_lower)
        pi_new = np.zeros((N + 1, NX))
        pi_new[N] = (W[N] @ v[N])[NU:]
        for k in range(N - 1, 0, -1):
            pi_new[k] = (W[k] @ v[k] + g[k] - rowgrad[k])[NU:] + A[k][:, NU:].T @ pi_new[k + 1]
        z[:N] += v[:N]; z[N, NU:] += v[N, NU:]
        pi = pi_new; lamh = lam_gen
    pobj = sum(pb.dt * O.stage_cost(pb, z[k], params[k])[0] for k in range(N))
    return z[:, NU:].copy(), z[:N, :NU].copy(), pobj, sizes
