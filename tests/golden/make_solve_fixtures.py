"""Writes tests/golden/solve_fixtures.json: whole-solve fixtures (inputs + outputs) for every BASELINE shape, so that the
[UPSTREAM] assumptions U1-U9 of DESIGN.md can be closed by anyone with an acados installation (tools/acados_replay.py replays
the inputs through the reference's own generated solver and compares).  Per trajectory:
  inputs   xinit [nx], x0 [(N+1) nvar] (AcadosParameters::x0 layout), params [N npar] (all_parameters layout)
  outputs  oracle at the reference's qp_tol = 1e-5, oracle at qp_tol = 1e-8 (the tightest tolerance above the float64 noise floor of the
           interior-point method on these QPs: profiles/round6_tight_tolerance_study.json) and 1e-9, and the independent active-set RTI
           (tests/independent_rti.py): xtraj, utraj, pobj, exit codes / iteration counts
Run from the repository root:  python tests/golden/make_solve_fixtures.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import independent_rti as I  # noqa: E402
import oracle_lib as O  # noqa: E402
from mpc_planner_amd import scenes  # noqa: E402

CASES = {
    "cfg1": (dict(N=20, M=4, B=1, guidance=False), dict(N=20, S=5, n_lin=0, M=4), [(90, 0), (91, 0)],
             "generate_jackalsimulator_solver.py configuration_basic-like stack: MPCBase + Contouring + EllipsoidConstraints(4)"),
    "cfg2": (dict(N=20, M=8, B=64), dict(N=20, S=5, n_lin=8, M=8), [(90, 5), (90, 40)],
             "configuration_tmpc: MPCBase + Contouring + GuidanceConstraints(EllipsoidConstraints), max_obstacles 8"),
    "cfg3": (dict(N=30, M=8, B=16, slack=True, n_decomp=12), dict(N=30, S=5, n_lin=8, M=8, n_slk=12, slack=1), [(90, 3)],
             "generate_rosnavigation_solver.py configuration_tmpc: slack model, + DecompConstraints(12)"),
    "cfg4": (dict(N=20, M=12, B=16, tmpc_pp=True), dict(N=20, S=5, n_lin=12, M=12), [(90, 2), (90, 16)],
             "configuration_tmpc with max_obstacles 12; trajectory 16 is the non-guided T-MPC++ planner (dummy topology rows)"),
    "cfg5": (dict(N=20, M=8, B=16, slack=True, n_scenario=24), dict(N=20, S=5, n_lin=0, M=0, n_slk=24, slack=1), [(90, 7)],
             "configuration_safe_horizon: slack model, ScenarioConstraints(24)"),
}

out = {"about": __doc__, "layout": "xinit [nx]; x0 [(N+1)*nvar] = [u_k; x_k] per node; params [N*npar] row k = stage k "
       "(acados_solver_interface.h:53-56); nvar = 7 (a, w | x, y, psi, v, spline) or 8 with the slack state last", "cases": []}
for name, (skw, pkw, picks, what) in CASES.items():
    pb = O.problem(**pkw); tight = O.problem(qp_tol=1e-9, **pkw); tight8 = O.problem(qp_tol=1e-8, **pkw)
    for scene, b in picks:
        for scene in range(scene, scene + 40):            # first scene from the nominal one whose pick is a full-length success
            sc = scenes.make_scene(scene, **skw)
            xi, x0, pa = sc["xinit"][b], sc["x0"][b], sc["params"][b]
            xo, uo, io = O.solve(pb, xi, x0, pa)
            if io.exit_code == 1 and io.sqp_iter == pb.n_sqp:
                break
        xt, ut, it_ = O.solve(tight, xi, x0, pa)
        x8, u8, i8 = O.solve(tight8, xi, x0, pa)
        print(name, scene, b, "1e-8:", i8.exit_code, i8.sqp_iter, "1e-9:", it_.exit_code, it_.sqp_iter, "max |x(1e-8) - x(1e-9)|", np.abs(x8 - xt).max())
        assert i8.exit_code == 1 and i8.sqp_iter == pb.n_sqp and np.abs(x8 - xt).max() < 1e-5
        assert io.exit_code == 1 and io.sqp_iter == pb.n_sqp, (name, scene, b, io.exit_code, io.sqp_iter)
        xa, ua, pobj_a, _ = I.rti_solve(pb, xi, x0, pa)
        assert np.abs(xa - xt).max() < 1e-6
        out["cases"].append(dict(
            config=name, modules=what, scene=scene, trajectory=b, N=pb.N, npar=pb.npar, nx=pb.nxe, nvar=pb.nve, problem=pkw,
            parameter_map={k: int(v) for k, v in sc["pm"]._params.items()} if hasattr(sc["pm"], "_params") else None,
            xinit=xi.tolist(), x0=x0.ravel().tolist(), params=pa.ravel().tolist(),
            oracle_qp_tol_1e_5=dict(xtraj=xo.ravel().tolist(), utraj=uo.ravel().tolist(), pobj=io.pobj, exit_code=io.exit_code,
                                    sqp_iter=io.sqp_iter, qp_iter_total=io.qp_iter_total, res_eq=io.res_eq),
            oracle_qp_tol_1e_8=dict(xtraj=x8.ravel().tolist(), utraj=u8.ravel().tolist(), pobj=i8.pobj, exit_code=i8.exit_code, sqp_iter=i8.sqp_iter,
                                    qp_iter_total=i8.qp_iter_total),
            oracle_qp_tol_1e_9=dict(xtraj=xt.ravel().tolist(), utraj=ut.ravel().tolist(), pobj=it_.pobj),
            active_set_rti=dict(xtraj=xa.ravel().tolist(), utraj=ua.ravel().tolist(), pobj=float(pobj_a))))
with open(os.path.join(HERE, "solve_fixtures.json"), "w") as fh:
    json.dump(out, fh)
print(len(out["cases"]), "cases,", os.path.getsize(os.path.join(HERE, "solve_fixtures.json")) // 1024, "KiB")
