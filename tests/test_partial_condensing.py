"""Numerical prototype of the next kernel lever: the stage-structured Newton system of one interior-point
iteration solved by a Riccati recursion over DOUBLE stages (every second state eliminated: partial condensing with block size 2;
n_u = 4, n_x = 5, half as many sequential stages) gives the same step and the same multipliers as the plain recursion / the dense
KKT solve.  Pure numpy; nothing here is product code -- it pins the algebra the HIP port has to reproduce."""
import numpy as np

NU, NX, NV = 2, 5, 7


def _problem(rng, N):
    H = []
    for k in range(N):
        M = rng.normal(size=(NV, NV))
        H.append(M @ M.T + 0.5 * np.eye(NV))
    M = rng.normal(size=(NX, NX))
    HN = M @ M.T + 0.5 * np.eye(NX)
    g = rng.normal(size=(N, NV)); gN = rng.normal(size=NX)
    BA = rng.normal(scale=0.4, size=(N, NX, NV)); BA[:, :, NU:] += np.eye(NX)
    b = rng.normal(scale=0.1, size=(N, NX))
    return H, HN, g, gN, BA, b


def _dense_kkt(H, HN, g, gN, BA, b):
    """min sum 1/2 v'Hv + g'v + terminal  s.t.  x_{k+1} = [B A] v_k + b_k,  x_0 = 0;  v_k = (u_k, x_k).  Returns v [N][7], x_N, pi [N+1][5]
    (pi_k: multiplier of the constraint that defines x_k, sign such that stationarity reads H v + g + [B A]' pi_{k+1} - [0; pi_k] = 0)."""
    N = len(H)
    nz = N * NV + NX
    nc = (N + 1) * NX
    K = np.zeros((nz + nc, nz + nc)); r = np.zeros(nz + nc)
    for k in range(N):
        K[k * NV:(k + 1) * NV, k * NV:(k + 1) * NV] = H[k]; r[k * NV:(k + 1) * NV] = -g[k]
    K[N * NV:nz, N * NV:nz] = HN; r[N * NV:nz] = -gN
    xs = lambda k: slice(k * NV + NU, k * NV + NV) if k < N else slice(N * NV, nz)
    C = np.zeros((nc, nz)); d = np.zeros(nc)
    C[0:NX, xs(0)] = -np.eye(NX)                                          # -x_0 = 0
    for k in range(N):
        rows = slice((k + 1) * NX, (k + 2) * NX)
        C[rows, k * NV:(k + 1) * NV] = BA[k]; C[rows, xs(k + 1)] -= np.eye(NX); d[rows] = -b[k]
    K[:nz, nz:] = C.T; K[nz:, :nz] = C; r[nz:] = d
    sol = np.linalg.solve(K, r)
    v = sol[:N * NV].reshape(N, NV); xN = sol[N * NV:nz]; pi = sol[nz:].reshape(N + 1, NX)
    return v, xN, pi


def _riccati(H, HN, g, gN, BA, b, nu):
    """Plain backward / forward recursion for stages with nu inputs (v = (u, x), [B A] of width nu + 5), x_0 = 0."""
    N = len(H)
    P = HN.copy(); p = gN.copy()
    Ks, ks, Ps, ps = [None] * N, [None] * N, [None] * (N + 1), [None] * (N + 1)
    Ps[N], ps[N] = P, p
    for k in range(N - 1, -1, -1):
        F = H[k] + BA[k].T @ P @ BA[k]
        f = g[k] + BA[k].T @ (P @ b[k] + p)
        Fuu, Fux, Fxx = F[:nu, :nu], F[:nu, nu:], F[nu:, nu:]
        Ks[k] = -np.linalg.solve(Fuu, Fux); ks[k] = -np.linalg.solve(Fuu, f[:nu])
        P = Fxx + Fux.T @ Ks[k]; p = f[nu:] + Fux.T @ ks[k]
        P = 0.5 * (P + P.T)
        Ps[k], ps[k] = P, p
    x = np.zeros(NX); v = []
    for k in range(N):
        u = Ks[k] @ x + ks[k]
        v.append(np.concatenate([u, x]))
        x = BA[k] @ v[-1] + b[k]
    return np.array(v), x, Ps, ps


def _condense_pairs(H, g, BA, b):
    """Stages (2j, 2j+1) -> one stage in y = (u_2j, u_2j+1, x_2j): cost Hessian / gradient, dynamics to x_{2j+2}, and the map back."""
    Hb, gb, BAb, bb, back = [], [], [], [], []
    for k in range(0, len(H), 2):
        E1 = np.zeros((NV, 9)); E1[:NU, :NU] = np.eye(NU); E1[NU:, 4:] = np.eye(NX)              # (u_k, x_k) = E1 y
        T = np.zeros((NV, 9)); T[:NU, NU:4] = np.eye(NU); T[NU:, :NU] = BA[k][:, :NU]; T[NU:, 4:] = BA[k][:, NU:]   # (u_k+1, x_k+1) = T y + t
        t = np.concatenate([np.zeros(NU), b[k]])
        Hb.append(E1.T @ H[k] @ E1 + T.T @ H[k + 1] @ T)
        gb.append(E1.T @ g[k] + T.T @ (g[k + 1] + H[k + 1] @ t))
        BAb.append(BA[k + 1] @ T)
        bb.append(BA[k + 1] @ t + b[k + 1])
        back.append((T, t))
    return Hb, np.array(gb), np.array(BAb), np.array(bb), back


def test_double_stage_recursion_gives_the_same_newton_step():
    rng = np.random.default_rng(3)
    for N in (2, 4, 20):
        H, HN, g, gN, BA, b = _problem(rng, N)
        v_ref, xN_ref, pi_ref = _dense_kkt(H, HN, g, gN, BA, b)
        v1, xN1, Ps, ps = _riccati(H, HN, g, gN, BA, b, NU)                      # what the kernels do today (in square-root form)
        assert np.abs(v1 - v_ref).max() < 1e-9 and np.abs(xN1 - xN_ref).max() < 1e-9
        Hb, gb, BAb, bb, back = _condense_pairs(H, g, BA, b)
        y, xN2, Pb, pb = _riccati(Hb, HN, gb, gN, BAb, bb, 2 * NU)
        assert np.abs(xN2 - xN_ref).max() < 1e-9
        # unpack: stage 2j from y directly, stage 2j+1 through the dynamics
        v2 = np.zeros_like(v_ref)
        for j, (T, t) in enumerate(back):
            v2[2 * j, :NU] = y[j, :NU]; v2[2 * j, NU:] = y[j, 4:]
            v2[2 * j + 1] = T @ y[j] + t
        assert np.abs(v2 - v_ref).max() < 1e-9
        # multipliers: pi at even nodes from the cost-to-go of the double-stage recursion, at odd nodes from stationarity in x_{2j+1}
        x_all = np.vstack([v2[:, NU:], xN2[None]])
        pi2 = np.zeros((N + 1, NX))
        for j in range(N // 2 + 1):
            pi2[2 * j] = Pb[j] @ x_all[2 * j] + pb[j] if j < N // 2 else HN @ xN2 + gN
        for j in range(N // 2):
            k = 2 * j + 1
            pi2[k] = (H[k] @ v2[k] + g[k] + BA[k].T @ pi2[k + 1])[NU:]
        # (the dense system's pi_0 belongs to the constraint x_0 = 0; the recursion's pi_k for k >= 1 are the dynamics multipliers)
        assert np.abs(pi2[1:] - pi_ref[1:]).max() < 1e-8


def test_chain_lengths():
    """What the restructuring buys on the sequential chain: pivots of the stage factorisations and stage hand-overs."""
    N = 20
    plain = dict(pivots=N * NV, stages=N)
    double = dict(pivots=(N // 2) * (2 * NU + NX), stages=N // 2)
    assert (plain["pivots"], double["pivots"]) == (140, 90) and double["stages"] * 2 == plain["stages"]
