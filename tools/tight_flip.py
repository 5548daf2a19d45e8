#!/usr/bin/env python3
"""Root cause of the integer disagreements at qp_tol = 1e-9 (round-5 verdict, next-3 (i)): BENCH_r05's `qp_tol_1e_9.parity` block had 1 of 128
trajectories ending differently on the device and in the oracle (oracle in the kernels' Riccati form).  For every trajectory of a sample whose
exit code / SQP count / interior-point count differs this script names

  * the trajectory (index in the bench launch), both sides' integers,
  * the SQP (RTI) iteration j* at which the interior-point counts first differ, and the interior-point iteration `it` of that QP at which one side
    stops and the other goes on -- i.e. WHICH stopping test of oracle/qp_ipm.c (`res_g <= tol && res_b <= tol && res_d <= tol && res_m <= tol`)
    flips, and on which of its four residuals,
  * that residual on both sides: the oracle's from its own trace (ORC_IPM_TRACE, %.17g), the device's by bisection on the tolerance -- the state
    after j* - 1 RTI iterations at 1e-9 is copied (tmpc_copy_state) into a handle with tolerance tol', ONE more iteration runs there, and the
    smallest tol' at which the QP stops at `it` is the device's worst residual at that iteration.

Run on the GPU box:  python tools/tight_flip.py [--scenes 512] [--sample 128] [--more 1920] --out profiles/round6_tight_flip.json
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_trace(O, pb, xi, x0, pr):
    """The oracle's residuals at every interior-point iteration of every QP of one solve: list over QPs of list of dicts."""
    fd_saved = os.dup(2)
    with tempfile.TemporaryFile(mode="w+b") as tf:
        os.environ["ORC_IPM_TRACE"] = "1"
        sys.stderr.flush()
        os.dup2(tf.fileno(), 2)
        try:
            xt, ut, info = O.solve(pb, xi, x0.ravel(), pr.ravel())
        finally:
            os.dup2(fd_saved, 2); os.close(fd_saved)
            del os.environ["ORC_IPM_TRACE"]
        tf.seek(0)
        text = tf.read().decode()
    qps = []
    for line in text.splitlines():
        w = line.split()
        if len(w) >= 12 and w[0] == "ipm" and w[1] == "it":
            it = int(w[2])
            if it == 0:
                qps.append([])
            qps[-1].append({"it": it, "res_g": float(w[4]), "res_b": float(w[6]), "res_d": float(w[8]), "res_m": float(w[10])})
    return qps, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=512)
    ap.add_argument("--sample", type=int, default=128, help="the bench's parity sample of the launch (np.linspace over the batch)")
    ap.add_argument("--more", type=int, default=1920, help="further evenly spread trajectories, for a rate")
    ap.add_argument("--qp-tol", type=float, default=1e-9)
    ap.add_argument("--form", type=int, default=0, choices=[0, 1],
                    help="Riccati form on BOTH sides: 0 = Schur complement (tmpc_dims.riccati_form 0, the oracle's riccati_form 1), 1 = square root (1 / 0)")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--max-cases", type=int, default=16, help="mismatching trajectories traced in detail (all are counted)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from mpc_planner_amd import scenes, solver
    import oracle_lib as O
    import bench
    wl = bench.WORKLOADS[a.workload]
    batch = scenes.make_batch(range(0, a.scenes), workers=bench.usable_cpus(), B=min(wl["traj"], 64), **wl["scene"])
    B = batch["xinit"].shape[0]
    idx_bench = np.unique(np.linspace(0, B - 1, min(a.sample, B)).round().astype(int))
    idx_more = np.unique(np.linspace(0, B - 1, min(a.more, B)).round().astype(int)) if a.more > 0 else np.zeros(0, int)
    idx = np.unique(np.concatenate([idx_bench, idx_more]))
    n = len(idx)
    tol = a.qp_tol
    dims = solver.default_dims(**wl["dims"], qp_tol=tol, riccati_form=a.form)
    xi, x0, pr = batch["xinit"][idx], batch["x0"][idx], batch["params"][idx]
    s = solver.BatchedSolver(dims, B_max=n)
    s.set_batch(xi, x0, pr); s.solve(); g = s.get(); info_kernel = s.kernel_info(); s.close()
    pb = O.problem(**wl.get("oracle_dims", wl["dims"]), qp_tol=tol, riccati_form=1 - a.form)
    xt, ut, o = O.solve_batch(pb, xi, x0.reshape(n, -1), pr.reshape(n, -1), num_threads=bench.usable_cpus())
    mism = np.where((g["exit_code"] != o["exit_code"]) | (g["sqp_iter"] != o["sqp_iter"]) | (g["qp_iter_total"] != o["qp_iter_total"]))[0]
    both = (g["exit_code"] == 1) & (o["exit_code"] == 1)
    agree = both & (g["qp_iter_total"] == o["qp_iter_total"])
    sx = np.maximum(np.abs(xt).max(axis=2, keepdims=True), 1.0)
    rel = (np.abs(g["xtraj"] - xt) / sx).max(axis=(1, 2))
    cases = []
    for m in mism[:a.max_cases]:
        c = {"trajectory_in_launch": int(idx[m]), "in_bench_parity_sample": bool(idx[m] in set(idx_bench.tolist())),
             "device": {k: int(g[k][m]) for k in ("exit_code", "qp_status", "sqp_iter", "qp_iter_total")},
             "oracle": {k: int(o[k][m]) for k in ("exit_code", "qp_status", "sqp_iter", "qp_iter_total")}}
        # ---- per RTI iteration: the device through the one-iteration protocol (bitwise the one-launch solve), the oracle from its trace ----
        one = solver.BatchedSolver(dims, B_max=1)
        one.set_batch(xi[m:m + 1], x0[m:m + 1], pr[m:m + 1])
        dev_it = []
        for j in range(dims.n_sqp):
            one.solve_iterations(1, keep_iterate=j > 0, keep_multipliers=j > 0, complete=True, new_solve=(j == 0))
            r1 = one.get()
            dev_it.append(int(r1["qp_iter_total"][0]))
            if int(r1["qp_status"][0]) != 0:
                break
        qps, _ = oracle_trace(O, pb, xi[m], x0[m], pr[m])
        orc_it = [q[-1]["it"] for q in qps]
        c["ipm_iterations_per_rti_iteration"] = {"device": dev_it, "oracle": orc_it}
        jstar = next((j for j in range(min(len(dev_it), len(orc_it))) if dev_it[j] != orc_it[j]), None)
        if jstar is None:
            c["first_difference"] = "the interior-point counts agree on every common RTI iteration; the loop lengths differ"
            one.close(); cases.append(c); continue
        it_dec = min(dev_it[jstar], orc_it[jstar])                  # the iteration at which one side's stopping test passed and the other's did not
        orc_res = qps[jstar][it_dec]
        worst_name = max(("res_g", "res_b", "res_d", "res_m"), key=lambda k: orc_res[k])
        # ---- the device's worst residual at (j*, it_dec): bisection on the tolerance of ONE iteration run from the copied state ----
        base = solver.BatchedSolver(dims, B_max=1)
        base.set_batch(xi[m:m + 1], x0[m:m + 1], pr[m:m + 1])
        if jstar > 0:
            base.solve_iterations(jstar, complete=True, new_solve=True)
        def count_at(t):
            d2 = solver.default_dims(**wl["dims"], qp_tol=float(t), riccati_form=a.form)
            h = solver.BatchedSolver(d2, B_max=1)
            h.set_batch(xi[m:m + 1], x0[m:m + 1], pr[m:m + 1])
            if jstar > 0:
                h.copy_state_from(base)
                h.solve_iterations(1, keep_iterate=True, keep_multipliers=True, complete=True, new_solve=True)
            else:
                h.solve_iterations(1, complete=True, new_solve=True)
            k = int(h.get()["qp_iter_total"][0]); h.close()
            return k
        lo, hi = tol * 1e-5, tol * 1e5
        ok_bracket = count_at(lo) > it_dec and count_at(hi) <= it_dec
        if ok_bracket:
            for _ in range(80):
                mid = (lo * hi) ** 0.5
                if count_at(mid) <= it_dec:
                    hi = mid
                else:
                    lo = mid
                if hi - lo <= 4e-16 * hi:
                    break
        base.close(); one.close()
        dev_worst = hi if ok_bracket else None
        c["first_difference"] = {
            "rti_iteration": int(jstar + 1), "ipm_iteration": int(it_dec),
            "stopping_test": "oracle/qp_ipm.c: res_g <= tol && res_b <= tol && res_d <= tol && res_m <= tol (the kernels: max of the four <= tol, csrc/tmpc_fast.hpp blk_residuals)",
            "deciding_residual": worst_name, "oracle_residuals": orc_res, "oracle_worst": orc_res[worst_name],
            "device_worst_by_bisection": dev_worst, "tolerance": tol,
            "oracle_stops_here": bool(orc_it[jstar] == it_dec), "device_stops_here": bool(dev_it[jstar] == it_dec),
            "relative_gap_of_the_two_residuals": (abs(dev_worst - orc_res[worst_name]) / tol) if dev_worst else None,
            "oracle_margin_to_tolerance_relative": (orc_res[worst_name] - tol) / tol,
            "device_margin_to_tolerance_relative": ((dev_worst - tol) / tol) if dev_worst else None}
        cases.append(c)
    kinds = {"exit_code": int((g["exit_code"] != o["exit_code"]).sum()), "sqp_iter": int((g["sqp_iter"] != o["sqp_iter"]).sum()),
             "qp_iter_total": int((g["qp_iter_total"] != o["qp_iter_total"]).sum()),
             "device_failed_oracle_ok": int(((g["exit_code"] != 1) & (o["exit_code"] == 1)).sum()), "oracle_failed_device_ok": int(((g["exit_code"] == 1) & (o["exit_code"] != 1)).sum()),
             "device_success_fraction": float((g["exit_code"] == 1).mean()), "oracle_success_fraction": float((o["exit_code"] == 1).mean())}
    out = {"what": "integer disagreements device <-> oracle (same Riccati form on both sides) at a tight QP tolerance, each traced to the stopping test that flips",
           "riccati_form": ["Schur complement (tmpc_dims.riccati_form 0 / oracle riccati_form 1)", "square root (tmpc_dims.riccati_form 1 / oracle riccati_form 0)"][a.form],
           "mismatch_kinds": kinds,
           "workload": f"{a.workload} bench launch ({a.scenes} scenes x {min(wl['traj'], 64)}), sample = the bench's {len(idx_bench)}-trajectory parity sample + {len(idx) - len(idx_bench)} more",
           "qp_tol": tol, "kernel": info_kernel, "library_sha256": bench.library_sha256(),
           "trajectories": int(n), "mismatching": int(len(mism)), "mismatching_in_bench_sample": int(sum(c["in_bench_parity_sample"] for c in cases)),
           "parity_max_rel_where_counts_agree": float(rel[agree].max()) if agree.any() else None,
           "parity_max_rel_where_both_succeed": float(rel[both].max()) if both.any() else None,
           "cases": cases}
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
