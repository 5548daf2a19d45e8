"""mpc_planner_amd -- MI355X-native batched SQP/NLP solve path for tud-amr/mpc_planner's T-MPC inner loop.

Host-side mirror of the reference's solver / module interface for this one hot path:
  parameters.py  -- parameter map (solver_generator/util/parameters.py, solver_definition.py:5-16)
  modules.py     -- per-stage parameter writers of the C++ modules (mpc_planner_modules/src/*.cpp)
  scenes.py      -- deterministic synthetic Jackal scenes
  solver.py      -- ctypes binding of the C-ABI (include/tmpc_hip.h) + batched optimize()
  csrc/          -- HIP kernels (gfx950) and the C-ABI shim
"""
__all__ = ["parameters", "modules", "scenes"]
