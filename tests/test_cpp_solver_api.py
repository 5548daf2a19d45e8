"""C++ host-side mirror of MPCPlanner::Solver over the C-ABI: generate the YAML maps + setSolverParameter* code,
compile tests/cpp/test_solver.cpp (written after the reference's mpc_planner_solver/test/test_solver.cpp) and run
its plumbing part on CPU; the solve part runs under -m gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "build", "generated_cfg2")
BIN = os.path.join(ROOT, "build", "test_solver")


def _build():
    import __graft_entry__ as g
    g.build()
    from mpc_planner_amd.generate_solver import generate_solver
    pm = generate_solver(GEN, N=20, max_obstacles=8, num_segments=5, guidance=True)
    assert pm.length() == 135
    cpp = os.path.join(ROOT, "mpc_planner_amd", "cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(cpp, "include"), "-I", os.path.join(GEN, "include"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_solver.cpp"),
           os.path.join(cpp, "src", "solver_interface.cpp"), os.path.join(GEN, "src", "mpc_planner_parameters.cpp"),
           "-L", os.path.join(ROOT, "mpc_planner_amd"), "-ltmpc_hip", "-Wl,-rpath," + os.path.join(ROOT, "mpc_planner_amd"),
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-o", BIN]
    subprocess.check_call(cmd)


def test_generated_maps_match_reference_layout():
    _build()
    pm = dict(l.strip().split(": ") for l in open(os.path.join(GEN, "config", "parameter_map.yaml")))
    assert pm["acceleration"] == "0" and pm["spline_x0_a"] == "8" and pm["lin_constraint_0_a1"] == "53"
    assert pm["ego_disc_radius"] == "77" and pm["ellipsoid_obst_7_r"] == "134" and pm["num parameters"] == "135"
    mm = open(os.path.join(GEN, "config", "model_map.yaml")).read()
    assert "a: [u, 0, -2.0, 2.0]" in mm and "spline: [x, 6, -1.0, 10000.0]" in mm


def test_cpp_solver_plumbing():
    _build()
    out = subprocess.run([BIN, os.path.join(GEN, "config")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "plumbing ok" in out.stdout, out.stdout + out.stderr


GEN_SLACK = os.path.join(ROOT, "build", "generated_cfg5")
BIN_SLACK = os.path.join(ROOT, "build", "test_solver_slack")


def _build_slack():
    """configuration_safe_horizon (cfg 5): slack model, 24 scenario halfspaces, no ellipsoid / topology rows."""
    import __graft_entry__ as g
    g.build()
    from mpc_planner_amd.generate_solver import generate_solver
    pm = generate_solver(GEN_SLACK, N=20, max_obstacles=8, num_segments=5, guidance=False, slack=True, ellipsoids=False, n_scenario=24)
    assert pm.length() == 127
    cpp = os.path.join(ROOT, "mpc_planner_amd", "cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(cpp, "include"), "-I", os.path.join(GEN_SLACK, "include"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_solver_slack.cpp"),
                           os.path.join(cpp, "src", "solver_interface.cpp"), os.path.join(GEN_SLACK, "src", "mpc_planner_parameters.cpp"),
                           "-L", os.path.join(ROOT, "mpc_planner_amd"), "-ltmpc_hip", "-Wl,-rpath," + os.path.join(ROOT, "mpc_planner_amd"),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-o", BIN_SLACK])


def test_cpp_solver_plumbing_slack_model():
    _build_slack()
    mm = open(os.path.join(GEN_SLACK, "config", "model_map.yaml")).read()
    assert "slack: [x, 7, 0.0, 5000.0]" in mm
    out = subprocess.run([BIN_SLACK, os.path.join(GEN_SLACK, "config")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "plumbing ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_solver_solves_slack_model():
    """Round-1 advisor finding (high): the slack model's ninth model_map entry overflowed tmpc_dims::lb/ub and tmpc_create failed."""
    if not os.path.exists(BIN_SLACK) or os.path.getmtime(BIN_SLACK) < os.path.getmtime(os.path.join(ROOT, "mpc_planner_amd", "libtmpc_hip.so")):
        _build_slack()
    out = subprocess.run([BIN_SLACK, os.path.join(GEN_SLACK, "config"), "--solve"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "solve ok" in out.stdout, out.stdout + out.stderr


def _stale():
    if not os.path.exists(BIN):
        return True
    cpp = os.path.join(ROOT, "mpc_planner_amd", "cpp")
    deps = [os.path.join(ROOT, "include", "tmpc_hip.h"), os.path.join(cpp, "include", "mpc_planner_solver", "solver_interface.h"),
            os.path.join(cpp, "src", "solver_interface.cpp"), os.path.join(ROOT, "tests", "cpp", "test_solver.cpp"),
            os.path.join(ROOT, "mpc_planner_amd", "libtmpc_hip.so")]
    return any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps)


@pytest.mark.gpu
def test_cpp_solver_solve_and_batch():
    if _stale():
        _build()
    out = subprocess.run([BIN, os.path.join(GEN, "config"), "--solve"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "solve ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_solver_over_a_generated_library():
    """The C++ Solver mirror linked against a per-configuration library produced by generate_solver_from_modules (T-MPC stack
    from plugin modules): same test program, same parameter names, plumbing + solve + batch."""
    from mpc_planner_amd.generate_solver import generate_solver_from_modules
    from mpc_planner_amd.codegen import stacks
    out = os.path.join(ROOT, "build", "generated_from_modules")
    st = stacks.settings(N=20, max_obstacles=8); st["integrator_step"] = 0.2
    lib = os.path.join(out, "lib", "libtmpc_hip_cpp_tmpc.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(
            os.path.getmtime(os.path.join(ROOT, "mpc_planner_amd", "csrc", f)) for f in os.listdir(os.path.join(ROOT, "mpc_planner_amd", "csrc"))):
        model, mm = stacks.tmpc(st)
        lib, meta = generate_solver_from_modules(out, "cpp_tmpc", mm, model, st)
        assert meta["npar"] == 135
    cpp = os.path.join(ROOT, "mpc_planner_amd", "cpp")
    exe = os.path.join(out, "test_solver_generated")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(cpp, "include"), "-I", os.path.join(out, "include"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_solver.cpp"),
                           os.path.join(cpp, "src", "solver_interface.cpp"), os.path.join(out, "src", "mpc_planner_parameters.cpp"),
                           lib, "-Wl,-rpath," + os.path.dirname(lib), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-o", exe])
    res = subprocess.run([exe, os.path.join(out, "config"), "--solve"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "solve ok" in res.stdout, res.stdout + res.stderr


BIN_SHARDED = os.path.join(ROOT, "build", "test_sharded")


def _build_sharded():
    import __graft_entry__ as g
    g.build()
    cpp = os.path.join(ROOT, "mpc_planner_amd", "cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(cpp, "include"), "-I", os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "test_sharded.cpp"), os.path.join(cpp, "src", "sharded_batch.cpp"),
                           "-L", os.path.join(ROOT, "mpc_planner_amd"), "-ltmpc_hip", "-Wl,-rpath," + os.path.join(ROOT, "mpc_planner_amd"),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lrccl", "-o", BIN_SHARDED])


def test_cpp_sharded_selection_compiles():
    """Native multi-GPU path (RCCL all-gather of the records + gathered FindBestPlanner) builds against librccl."""
    _build_sharded()


@pytest.mark.gpu
def test_cpp_sharded_selection_one_rank(tmp_path):
    import numpy as np
    from mpc_planner_amd import scenes
    if not os.path.exists(BIN_SHARDED) or os.path.getmtime(BIN_SHARDED) < os.path.getmtime(os.path.join(ROOT, "mpc_planner_amd", "libtmpc_hip.so")):
        _build_sharded()
    sc = scenes.make_batch(range(3, 6), N=20, M=8, B=16)
    B = sc["xinit"].shape[0]
    f = str(tmp_path / "batch.bin")
    np.concatenate([[B, 20, 3], sc["xinit"].ravel(), sc["x0"].ravel(), sc["params"].ravel()]).astype(float).tofile(f)
    out = subprocess.run([BIN_SHARDED, f], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sharded ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
