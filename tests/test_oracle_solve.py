"""CPU oracle, whole-solve checks (the reference pins no solve result -- SURVEY 8c -- so these are
self-validation: Lagrangian-Hessian assembly vs finite differences, QP optimality, converged-SQP optimum
vs scipy SLSQP on the same NLP, selection semantics)."""
import numpy as np
import pytest
import scipy.optimize as so

import oracle_lib as O
from mpc_planner_amd import scenes

N, M, S = 20, 8, 5


def _scene(idx=1, B=8):
    sc = scenes.make_scene(idx, N=N, M=M, B=B)
    return sc, O.problem(N=N, S=S, n_lin=M, M=M)


def test_lagrangian_hessian_assembly_matches_finite_differences():
    sc, pb = _scene(1, B=4)
    b = 2
    xt, ut, info, dbg = O.solve(pb, sc["xinit"][b], sc["x0"][b], sc["params"][b], debug_iter=3)
    Wraw = np.array(dbg.W_raw).reshape(-1, 7, 7); zin = np.array(dbg.z_in).reshape(-1, 7)
    pi = np.array(dbg.pi_in).reshape(-1, 5); lamh = np.array(dbg.lamh_in).reshape(-1, O.MAX_NH)
    assert np.abs(lamh).max() > 1e-6 and np.abs(pi).max() > 1e-6     # multipliers are live at iteration 3

    def grad_L(k, z):
        _, g, _ = O.stage_cost(pb, z, sc["params"][b, k])
        _, J, _ = O.discrete_dynamics(pb, z)
        _, D, _ = O.stage_constraints(pb, z, sc["params"][b, k])
        return pb.dt * g + J.T @ pi[k + 1] + D.T @ lamh[k, :2 * M]

    for k in (0, 5, 12, 19):
        H = np.zeros((7, 7)); eps = 1e-6
        for j in range(7):
            e = np.zeros(7); e[j] = eps
            H[:, j] = (grad_L(k, zin[k] + e) - grad_L(k, zin[k] - e)) / (2 * eps)
        np.testing.assert_allclose(Wraw[k], H, rtol=2e-6, atol=2e-7)
    # MIRROR output is SPD with eigenvalues >= eps
    W = np.array(dbg.W).reshape(-1, 7, 7)
    for k in range(N):
        assert np.linalg.eigvalsh(W[k]).min() >= 1e-4 * (1 - 1e-9)


def test_qp_solution_satisfies_linearised_dynamics_and_rows():
    sc, pb = _scene(1, B=4)
    b = 1
    xt, ut, info, dbg = O.solve(pb, sc["xinit"][b], sc["x0"][b], sc["params"][b], debug_iter=0)
    dz = np.array(dbg.dz).reshape(-1, 7)[:N + 1]; BA = np.array(dbg.BA).reshape(-1, 5, 7)
    bb = np.array(dbg.b).reshape(-1, 5); h = np.array(dbg.h).reshape(-1, O.MAX_NH); D = np.array(dbg.D).reshape(-1, O.MAX_NH, 7)
    for k in range(N):
        np.testing.assert_allclose(dz[k + 1, 2:], BA[k] @ dz[k] + bb[k], atol=2e-5)
        hl = h[k, :2 * M] + D[k, :2 * M] @ dz[k]
        assert (hl[:M] <= 2e-5).all() and (hl[M:] >= 1 - 2e-5).all()
    np.testing.assert_allclose(dz[0, 2:], sc["xinit"][b] - sc["x0"][b][0, 2:], atol=1e-12)


def _nlp_functions(pb, params, xinit):
    nz = N * 7 + 5

    def unpack(w):
        return [w[7 * k:7 * k + 7] for k in range(N)] + [np.concatenate([[0, 0], w[7 * N:]])]

    def f(w):
        zs = unpack(w)
        return sum(pb.dt * O.stage_cost(pb, zs[k], params[k])[0] for k in range(N))

    def fg(w):
        zs = unpack(w); g = np.zeros(nz)
        for k in range(N):
            g[7 * k:7 * k + 7] = pb.dt * O.stage_cost(pb, zs[k], params[k])[1]
        return g

    def ceq(w):
        zs = unpack(w); out = [zs[0][2:] - xinit]
        for k in range(N):
            out.append(O.discrete_dynamics(pb, zs[k])[0] - zs[k + 1][2:])
        return np.concatenate(out)

    def cineq(w):  # >= 0
        zs = unpack(w); out = []
        for k in range(N):
            h = O.stage_constraints(pb, zs[k], params[k])[0]
            out.append(-h[:M]); out.append(h[M:] - 1.0)
        return np.concatenate(out)

    lb = np.concatenate([np.array(pb.lb)] * N + [np.full(5, -np.inf)])
    ub = np.concatenate([np.array(pb.ub)] * N + [np.full(5, np.inf)])
    lb[2:7] = -np.inf; ub[2:7] = np.inf        # x_0 is fixed by the equality, not boxed
    return f, fg, ceq, cineq, lb, ub


@pytest.mark.parametrize("scene,b", [(1, 3), (5, 20)])
def test_converged_sqp_matches_scipy_slsqp(scene, b):
    """Run the oracle as a converged SQP (many RTI iterations, tight QP tolerance) and check that an
    independent NLP solver started there cannot move: same optimum, same objective."""
    sc = scenes.make_scene(scene, N=N, M=M, B=64)
    pb = O.problem(N=N, S=S, n_lin=M, M=M, n_sqp=60, qp_tol=1e-8)
    xt, ut, info = O.solve(pb, sc["xinit"][b], sc["x0"][b], sc["params"][b])
    assert info.exit_code == 1 and info.res_eq < 1e-9
    w0 = np.concatenate([np.concatenate([ut, xt[:N]], 1).ravel(), xt[N]])
    f, fg, ceq, cineq, lb, ub = _nlp_functions(pb, sc["params"][b], sc["xinit"][b])
    assert abs(f(w0) - info.pobj) < 1e-12
    assert np.abs(ceq(w0)).max() < 1e-9 and cineq(w0).min() > -1e-8
    res = so.minimize(f, w0, jac=fg, method="SLSQP", bounds=list(zip(lb, ub)),
                      constraints=[dict(type="eq", fun=ceq), dict(type="ineq", fun=cineq)],
                      options=dict(maxiter=200, ftol=1e-14))
    assert res.fun >= info.pobj - 1e-7           # scipy cannot improve on the oracle's optimum
    assert abs(res.fun - info.pobj) < 1e-6
    assert np.abs(res.x - w0).max() < 5e-3


def test_rti_iterations_approach_the_converged_optimum():
    sc = scenes.make_scene(1, N=N, M=M, B=8)
    pb10 = O.problem(N=N, S=S, n_lin=M, M=M)
    pbc = O.problem(N=N, S=S, n_lin=M, M=M, n_sqp=60, qp_tol=1e-8)
    x10, u10, i10 = O.solve(pb10, sc["xinit"][2], sc["x0"][2], sc["params"][2])
    xc, uc, ic = O.solve(pbc, sc["xinit"][2], sc["x0"][2], sc["params"][2])
    assert i10.sqp_iter == 10 and i10.exit_code == 1
    assert np.abs(x10 - xc).max() < 5e-2 and abs(i10.pobj - ic.pobj) < 1e-2


def test_find_best_planner_semantics():
    """FindBestPlanner (guidance_constraints.cpp:416-434): strict '<' -> lowest index on ties,
    failures and disabled planners skipped, -1 if none, init 1e10."""
    obj = np.array([3.0, 1.0, 1.0, 0.5, 2.0]); ec = np.array([1, 1, 1, 4, 1], np.int32)
    assert O.find_best(obj, ec) == 1
    assert O.find_best(obj, ec, disabled=[0, 1, 0, 0, 0]) == 2
    assert O.find_best(obj, np.zeros(5, np.int32)) == -1
    assert O.find_best(np.full(3, 2e10), np.ones(3, np.int32)) == -1


def test_infeasible_guess_reports_qp_failure():
    sc = scenes.make_scene(2, N=N, M=M, B=64)
    pm = sc["pm"]
    for b in (3, 10, 40):               # contradictory topology rows: x <= x_k - 5 and x >= x_k + 5 (no feasible QP point)
        for j, sg in ((0, 1.0), (1, -1.0)):
            sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_a1")] = sg
            sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_a2")] = 0.0
            sc["params"][b, 1:, pm.index(f"lin_constraint_{j}_b")] = sg * sc["x0"][b, 1:-1, 2] - 5.0
    pb = O.problem(N=N, S=S, n_lin=M, M=M)
    xt, ut, info = O.solve_batch(pb, sc["xinit"], sc["x0"].reshape(64, -1), sc["params"].reshape(64, -1))
    assert (info["exit_code"][[3, 10, 40]] == 4).all() and (info["exit_code"] == 1).sum() >= 50
    bad = info["exit_code"] == 4
    assert (info["qp_status"][bad] != 0).all() or (info["res_eq"][bad] > 1e-2).all()
