#!/usr/bin/env python3
"""Replay tests/golden/solve_fixtures.json through the REFERENCE's own acados solver (closes DESIGN.md's U1-U9 on a machine that
has what this build machine lacks: acados + acados_template + casadi, and a checkout of tud-amr/mpc_planner).

    export ACADOS_SOURCE_DIR=...; export LD_LIBRARY_PATH=$ACADOS_SOURCE_DIR/lib
    python tools/acados_replay.py /path/to/mpc_planner [--qp-tol 1e-8] [cfg2 cfg1 ...]

For every fixture of the requested configurations it builds the reference's module stack with the reference's own scripts
(mpc_planner_jackalsimulator/scripts/generate_jackalsimulator_solver.py: configuration_basic / configuration_tmpc /
configuration_safe_horizon; rosnavigation's configuration_tmpc for cfg 3), generates the acados solver exactly as
solver_generator/generate_acados_solver.py:generate_acados_solver does, and then drives it the way
mpc_planner_solver/src/acados_solver_interface.cpp does:
    lbx_0 = ubx_0 = xinit (:124-125); p_k = all_parameters[k] (k = N reuses row N-1, :127-135); x/u warm start (:274-284);
    rti_phase 0; n_sqp x solve() with the loop exit on qp_status != 0 (:99-117); cost via get_cost(); x, u out (:171-174).
It prints, per fixture, max relative per-stage differences to the stored outputs (oracle at qp_tol 1e-5 / 1e-8 / 1e-9 and the
active-set RTI).  Expected if U1-U9 hold: differences of the order the stored outputs have among themselves (<= ~1e-3 at
qp_tol 1e-5, see tests/test_independent_rti.py); a structural disagreement (terminal cost, h at node 0, bounds at node N,
stage-cost scaling) shows up as 1e-2 or more and names the assumption to fix.
It ends with a PASS / FAIL table at the tolerance that is certifiable for the chosen QP tolerance (round-3 verdict item 5):
    --qp-tol 1e-8   (overrides the generator's qp_tol = 1e-5: the iterate no longer depends on how the QP solver reaches its tolerance)
                    PASS <=> max relative per-stage difference to `oracle_qp_tol_1e_8` and `active_set_rti` <= 1e-5 -- the north star's 1e-4 with a
                    decade of margin; what this repository observes between its own independent implementations there is <= 2e-6.
                    USE 1e-8, NOT 1e-9 (round 6, profiles/round6_tight_tolerance_study.json): at 1e-9 a float64 interior-point method is below the
                    noise floor of its own stationarity residual on these QPs -- two correct implementations then disagree on iteration counts in
                    0.6 % (cfg 2) to 13 % (cfg 3) of the solves and 0.5 - 13 % of the solves break down, whatever the Riccati form; at 1e-8 this
                    repository's two implementations agree on every integer in 2042 of 2042 cfg 2 solves.  (--qp-tol 1e-9 is still accepted and
                    compares with `oracle_qp_tol_1e_9`.)  The Riccati form HPIPM runs (square_root_alg = 1 in its default mode) is available on
                    the device as tmpc_dims.riccati_form = 1 and is the oracle's default; at 1e-8 the two forms give the same iterates.
    default 1e-5    PASS <=> difference to `oracle_qp_tol_1e_5` <= 2e-3: the spread that correct implementations of the reference's configuration
                    show among themselves at that tolerance (profiles/round4_c_iterate_spread.json: 1e-4 .. 1e-3 on a few per cent of the
                    trajectories); a pass here says "no structural disagreement (U1-U9)", not "1e-4".
NOT run in this repository's CI: none of acados / casadi / the reference checkout exist on the build or GPU machines."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(os.path.dirname(HERE), "tests", "golden", "solve_fixtures.json")


QP_TOL = None            # --qp-tol: overrides the reference generator's qp_tol (generate_acados_solver.py:162) for the tight-tolerance check


def build_reference_solver(ref, case):
    sys.path[:0] = [os.path.join(ref, "solver_generator"), os.path.join(ref, "mpc_planner_modules", "scripts")]
    import yaml
    from generate_acados_solver import generate_acados_solver
    cfg, N = case["config"], case["N"]
    pkg = "mpc_planner_rosnavigation" if cfg == "cfg3" else "mpc_planner_jackalsimulator"
    sys.path.insert(0, os.path.join(ref, pkg, "scripts"))
    gen = __import__("generate_rosnavigation_solver" if cfg == "cfg3" else "generate_jackalsimulator_solver")
    with open(os.path.join(ref, pkg, "config", "settings.yaml")) as fh:      # (util.files.load_settings resolves its path from sys.argv[0])
        settings = yaml.safe_load(fh)
    settings["N"] = N
    settings["integrator_step"] = 0.2
    settings["n_discs"] = 1
    settings["max_obstacles"] = {"cfg1": 4, "cfg2": 8, "cfg3": 8, "cfg4": 12, "cfg5": 8}[cfg]
    settings["contouring"]["num_segments"] = 5
    settings["contouring"]["dynamic_velocity_reference"] = False
    settings["solver_settings"]["solver"] = "acados"
    settings["solver_settings"]["acados"]["solver_type"] = "SQP_RTI"
    settings["name"] = "replay_" + cfg
    stack = {"cfg1": "configuration_basic", "cfg2": "configuration_tmpc", "cfg3": "configuration_tmpc", "cfg4": "configuration_tmpc",
             "cfg5": "configuration_safe_horizon"}[cfg]
    model, modules = getattr(gen, stack)(settings)
    solver, _sim = generate_acados_solver(modules, settings, model, False)
    if QP_TOL is not None:                                         # the generator hard-codes qp_tol = 1e-5 (:162); the solver object takes the override
        for opt in ("qp_tol_stat", "qp_tol_eq", "qp_tol_ineq", "qp_tol_comp"):
            solver.options_set(opt, QP_TOL)
    pm = settings["params"]._params                                # util/parameters.py:13,44: name -> index
    for name, idx in (case["parameter_map"] or {}).items():       # the fixture's rows must be in the generator's own order
        if pm.get(name) != idx:
            raise SystemExit(f"parameter '{name}': fixture index {idx}, reference generator {pm.get(name)}")
    if settings["params"].length() != case["npar"]:
        raise SystemExit(f"npar: fixture {case['npar']}, reference generator {settings['params'].length()}")
    return solver


def replay(solver, case, n_sqp=10):
    N, nx, nvar, npar = case["N"], case["nx"], case["nvar"], case["npar"]
    nu = nvar - nx
    xinit = np.array(case["xinit"]); x0 = np.array(case["x0"]).reshape(N + 1, nvar); params = np.array(case["params"]).reshape(N, npar)
    solver.reset()
    solver.set(0, "lbx", xinit); solver.set(0, "ubx", xinit)
    for k in range(N + 1):
        solver.set(k, "p", params[min(k, N - 1)])
        solver.set(k, "x", x0[k, nu:])
        if k < N:
            solver.set(k, "u", x0[k, :nu])
    solver.options_set("rti_phase", 0)
    status = 0
    for _ in range(n_sqp):
        status = solver.solve()
        if solver.get_stats("qp_stat")[-1] != 0:
            break
    xt = np.array([solver.get(k, "x") for k in range(N + 1)]); ut = np.array([solver.get(k, "u") for k in range(N)])
    return xt, ut, float(solver.get_cost()), status


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    global QP_TOL
    args = sys.argv[1:]
    if "--qp-tol" in args:
        i = args.index("--qp-tol"); QP_TOL = float(args[i + 1]); del args[i:i + 2]
    ref = os.path.abspath(args[0])
    want = set(args[1:]) or None
    tight = QP_TOL is not None and QP_TOL <= 1e-8
    table = []
    cases = [c for c in json.load(open(FIX))["cases"] if want is None or c["config"] in want]
    solvers = {}
    for c in cases:
        if c["config"] not in solvers:
            solvers[c["config"]] = build_reference_solver(ref, c)
        xt, ut, cost, status = replay(solvers[c["config"]], c)
        N, nx = c["N"], c["nx"]
        line = [f"{c['config']} scene {c['scene']} trajectory {c['trajectory']}: acados status {status}, cost {cost:.9g}"]
        diff = {}
        for key in ("oracle_qp_tol_1e_5", "oracle_qp_tol_1e_8", "oracle_qp_tol_1e_9", "active_set_rti"):
            xr = np.array(c[key]["xtraj"]).reshape(N + 1, nx); ur = np.array(c[key]["utraj"]).reshape(N, 2)
            sx = np.maximum(np.abs(xr).max(axis=1, keepdims=True), 1.0); su = np.maximum(np.abs(ur).max(axis=1, keepdims=True), 1.0)
            diff[key] = max(float((np.abs(xt - xr) / sx).max()), float((np.abs(ut - ur) / su).max()))
            line.append(f"{key}: x {(np.abs(xt - xr) / sx).max():.2e} u {(np.abs(ut - ur) / su).max():.2e} "
                        f"cost {abs(cost - c[key]['pobj']) / max(1.0, abs(c[key]['pobj'])):.2e}")
        print(" | ".join(line))
        tight_key = "oracle_qp_tol_1e_9" if (QP_TOL is not None and QP_TOL < 5e-9) else "oracle_qp_tol_1e_8"
        worst = max(diff[tight_key], diff["active_set_rti"]) if tight else diff["oracle_qp_tol_1e_5"]
        table.append((f"{c['config']} scene {c['scene']} trajectory {c['trajectory']}", worst, worst <= (1e-5 if tight else 2e-3)))
    tol = 1e-5 if tight else 2e-3
    print(f"\n{'fixture':<40} {'max rel. per stage':>20}   verdict at {tol:g} ({'acados qp_tol ' + str(QP_TOL) + ': vs the oracle at the same tolerance and the active-set RTI' if tight else 'acados qp_tol 1e-5: vs oracle 1e-5'})")
    for name, w, ok in table:
        print(f"{name:<40} {w:>20.3e}   {'PASS' if ok else 'FAIL'}")
    n_fail = sum(not ok for _, _, ok in table)
    print(f"\n{'PASS' if n_fail == 0 else 'FAIL'}: {len(table) - n_fail} / {len(table)} fixtures within {tol:g}")
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
