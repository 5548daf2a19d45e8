"""Solver generation for user-defined cost / constraint modules.

The reference builds its NLP from python modules that write CasADi expressions (`mpc_planner_modules/scripts/*.py`
over `solver_generator/control_modules.py`) and lets CasADi + acados generate C for them
(`solver_generator/generate_acados_solver.py:27-65,190`).  This package is the MI355X-native counterpart:

  symbolic.py   a `casadi`-compatible facade (SX.sym, cos, sqrt, vertcat, ...) backed by sympy, so module scripts written
                against CasADi run unmodified (`install_as_casadi()`)
  plugin.py     the module protocol (ModuleManager / ObjectiveModule / ConstraintModule / Parameters / model) and the
                assembly rules of solver_definition.py:5-76
  library.py    a library of modules written against that protocol (MPC base weights, contouring, path reference velocity, goal, ellipsoids,
                topology halfspaces, decomp / scenario halfspaces, Gaussian chance constraints)
  emit.py       exact first / second derivatives, common-subexpression elimination and emission of the HIP stage
                functions (`tmpc_gen::cost`, `tmpc_gen::rows`) that `csrc/tmpc_stage.hpp` compiles into the solve kernel
  build.py      hipcc build of a per-configuration `libtmpc_hip_<name>.so` with the same C-ABI
  cpp_glue.py   modules.h / definitions.h / modules.cmake for the reference's C++ module classes (generate_cpp_files.py:11-95)
"""
