import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lanes: needs the optional lane-per-trajectory kernel family (csrc/tmpc_lanes.hip): runs only with TMPC_BUILD_LANES=1 "
                                       "in the environment (build() then builds and links the family and its CPU twin); skipped otherwise")


def pytest_collection_modifyitems(config, items):
    """The lane-per-trajectory family is an optional part of the library since round 5 (it loses to the wave kernels on every shape): its tests
    run when the build contains it."""
    if os.environ.get("TMPC_BUILD_LANES", "0") == "1":
        return
    skip = pytest.mark.skip(reason="optional lane-per-trajectory kernels not built (TMPC_BUILD_LANES=1 builds and tests them)")
    for item in items:
        if item.get_closest_marker("lanes") is not None:
            item.add_marker(skip)


import pytest


@pytest.fixture
def lab_library(monkeypatch):
    """Tests that have to reach ONE kernel family set TMPC_* kernel-selection overrides in the environment.  Only the lab build of the library reads them
    (mpc_planner_amd/libtmpc_hip_lab.so: the same kernel objects behind a C-ABI unit compiled with -DTMPC_LAB_SWITCHES); the product library ignores the
    environment.  This fixture makes the lab library the default of mpc_planner_amd.solver for the test."""
    from mpc_planner_amd import solver
    monkeypatch.setattr(solver, "LIB_PATH", solver.LAB_LIB_PATH)
    assert solver.load_library().tmpc_has_lab_switches() == 1
