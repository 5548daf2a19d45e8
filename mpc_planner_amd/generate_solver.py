"""Offline half of the drop-in: emit what the reference's solver generator emits for the C++ side
(solver_generator/generate_solver.py:34-56), but for the HIP backend -- no CasADi/acados code generation is
needed because the stage functions are hand-written HIP (csrc/tmpc_stage.hpp).

Outputs into an output directory (mirrors mpc_planner_solver/config + include/mpc_planner_solver):
  config/parameter_map.yaml   name -> index, "num parameters"            (util/parameters.py:70-76)
  config/model_map.yaml       name -> [x|u, index in z, lb, ub]          (solver_model.py:118-128)
  config/solver_settings.yaml N, nx, nu, nvar, npar                      (generate_solver.py:38-46)
  include/mpc_planner_solver/hip_solver_dims.h      SOLVER_* macros (stand-in for acados_solver_Solver.h)
  include/mpc_planner_solver/mpc_planner_parameters.h / src/mpc_planner_parameters.cpp
                              setSolverParameter<Bundle>(k, params, value, index)  (generate_cpp_files.py:204-260)
  include/mpc_planner_modules/{modules.h, definitions.h}, modules.cmake   (generate_solver_from_modules only;
                              codegen/cpp_glue.py after generate_cpp_files.py:11-95)
"""
import os

from .parameters import MODEL_MAP_UNICYCLE, MODEL_MAP_UNICYCLE_SLACK, define_parameters


def _yaml_dump_flat(d):
    lines = []
    for k, v in d.items():
        if isinstance(v, (list, tuple)):
            lines.append(f"{k}: [{', '.join(str(x) for x in v)}]")
        else:
            lines.append(f"{k}: {v}")
    return "\n".join(lines) + "\n"


def generate_solver(out_dir, N=20, max_obstacles=8, num_segments=5, guidance=True, n_sqp=10, dt=0.2, slack=False,
                    ellipsoids=True, n_scenario=0, n_decomp=0, curvature_aware=False, gaussian=False):
    """slack / n_scenario / n_decomp select the slack-model configurations (configuration_safe_horizon,
    generate_jackalsimulator_solver.py:67-90; rosnavigation configuration_tmpc, generate_rosnavigation_solver.py:86-108).
    curvature_aware: CurvatureAwareContouringModule instead of ContouringModule (same parameter map: its define_parameters adds
    contour, lag, terminal_*, the spline rows and leaves velocity / reference_velocity to MPCBaseModule,
    curvature_aware_contouring.py:22-46) -> SOLVER_COST_MODEL 1 (tmpc_dims::cost_model).
    gaussian: GaussianConstraintModule as the collision-avoidance module instead of the ellipsoids (mpc_planner_jackal's default,
    generate_jackal_solver.py:53-73; gaussian_constraints.py:40-52 parameter layout) -> SOLVER_ROW_MODEL 1 (tmpc_dims::row_model)."""
    pm = define_parameters(num_segments, max_obstacles, guidance=guidance, slack=slack, ellipsoids=ellipsoids and not gaussian,
                           n_scenario=n_scenario, n_decomp=n_decomp, gaussian=gaussian)
    return _write_host_side(out_dir, pm, N, slack, n_sqp, dt, n_lin=(max_obstacles if guidance else 0),
                            M=(max_obstacles if (ellipsoids or gaussian) else 0), n_slk=n_scenario + n_decomp, num_segments=num_segments,
                            max_obstacles=max_obstacles, cost_model=int(bool(curvature_aware)), row_model=int(bool(gaussian)))


def generate_solver_from_modules(out_dir, name, modules, model, settings, n_sqp=10):
    """The reference's generate_solver(modules, model, settings) (solver_generator/generate_solver.py:34-56) for an arbitrary
    module stack: emits and compiles the stage functions (mpc_planner_amd/codegen -> <out_dir>/lib/libtmpc_hip_<name>.so,
    same C-ABI) and writes the host side (YAML maps, setSolverParameter* functions, dims header) for the C++ `Solver`
    mirror, which then links that library instead of libtmpc_hip.so.  Returns (library path, meta)."""
    from .codegen import build
    lib, meta = build.build_generated_solver(name, modules, model, settings, os.path.join(out_dir, "lib"))
    from .codegen import plugin
    pm = plugin.Parameters(); plugin.define_parameters(modules, pm, settings)
    # in a generated library every row is one of the nh normalised rows; SOLVER_S only informs host code (segment loops)
    _write_host_side(out_dir, pm, settings["N"], bool(meta["slack"]), n_sqp, settings.get("integrator_step", 0.2), n_lin=meta["nh"], M=0,
                     n_slk=0, num_segments=settings.get("contouring", {}).get("num_segments", 1),
                     max_obstacles=settings.get("max_obstacles", 0), model=model)
    from .codegen import cpp_glue
    cpp_glue.write_module_glue(out_dir, modules)        # modules.h / definitions.h / modules.cmake for mpc_planner_modules
    return lib, meta


def _model_map(slack, model=None):
    """model_map.yaml rows `name: [x|u, index, lb, ub]` (solver_model.py:118-128).  A plugin model's own bounds (model.set_bounds,
    solver_model.py:107-116) are written out, as the reference wires them into lbx/ubx/lbu/ubu (generate_acados_solver.py:100-107);
    the C++ Solver mirror hands them to tmpc_create through tmpc_dims::lb/ub."""
    base = MODEL_MAP_UNICYCLE_SLACK if slack else MODEL_MAP_UNICYCLE
    if model is None or not hasattr(model, "lower_bound"):
        return base
    out = type(base)()
    names = list(getattr(model, "inputs", [])) + list(getattr(model, "states", []))
    for name, (kind, idx, lb, ub) in base.items():
        if names and name not in names:
            continue            # a model without this variable (SecondOrderUnicycleModel has no `spline`: solver_model.py:170-191): its model_map.yaml has no
                                # such row -- module code that loops over the map never names the kernels' inert padding slot; strides stay nx = 5, nvar = 7
        if idx < len(model.lower_bound):
            lb, ub = float(model.lower_bound[idx]), float(model.upper_bound[idx])
        out[name] = [kind, idx, lb, ub]
    return out


def _write_host_side(out_dir, pm, N, slack, n_sqp, dt, n_lin, M, n_slk, num_segments, max_obstacles, model=None, cost_model=0, row_model=0):
    npar = pm.length()
    nu, nx = 2, 5 + int(bool(slack))
    cfg = os.path.join(out_dir, "config"); inc = os.path.join(out_dir, "include", "mpc_planner_solver")
    src = os.path.join(out_dir, "src")
    for p in (cfg, inc, src):
        os.makedirs(p, exist_ok=True)
    open(os.path.join(cfg, "parameter_map.yaml"), "w").write(_yaml_dump_flat(pm.as_dict()))
    open(os.path.join(cfg, "model_map.yaml"), "w").write(_yaml_dump_flat(_model_map(slack, model)))
    open(os.path.join(cfg, "solver_settings.yaml"), "w").write(_yaml_dump_flat(
        dict(N=N, nx=nx, nu=nu, nvar=nx + nu, npar=npar, integrator_step=dt, iterations=n_sqp,
             n_lin=n_lin, max_obstacles=max_obstacles, num_segments=num_segments)))
    with open(os.path.join(inc, "hip_solver_dims.h"), "w") as f:
        f.write("/** autogenerated by mpc_planner_amd.generate_solver (stand-in for acados_solver_Solver.h) */\n"
                "#ifndef HIP_SOLVER_DIMS_H\n#define HIP_SOLVER_DIMS_H\n"
                f"#define SOLVER_N {N}\n#define SOLVER_NX {nx}\n#define SOLVER_NU {nu}\n#define SOLVER_NP {npar}\n"
                f"#define SOLVER_NLIN {n_lin}\n#define SOLVER_M {M}\n"
                f"#define SOLVER_NSLK {n_slk}\n#define SOLVER_SLACK {int(bool(slack))}\n#define SOLVER_MAX_OBSTACLES {max_obstacles}\n"
                f"#define SOLVER_S {num_segments}\n#define SOLVER_NSQP {n_sqp}\n#define SOLVER_DT {dt}\n#define SOLVER_COST_MODEL {cost_model}\n#define SOLVER_ROW_MODEL {row_model}\n#endif\n")
    with open(os.path.join(inc, "mpc_planner_parameters.h"), "w") as h, \
            open(os.path.join(src, "mpc_planner_parameters.cpp"), "w") as c:
        h.write("/** autogenerated by mpc_planner_amd.generate_solver */\n#ifndef __MPC_PLANNER_PARAMETERS_H__\n"
                "#define __MPC_PLANNER_PARAMETERS_H__\n\nnamespace MPCPlanner{\n\nstruct AcadosParameters;\n")
        c.write("#include <mpc_planner_solver/mpc_planner_parameters.h>\n\n#include <mpc_planner_solver/solver_interface.h>\n"
                "namespace MPCPlanner{\n\n")
        for key, indices in pm.parameter_bundles.items():
            fn = key.replace("_", " ").title().replace(" ", "")
            if len(indices) == 1:
                h.write(f"void setSolverParameter{fn}(int k, AcadosParameters& params, const double value, int index=0);\n")
                c.write(f"void setSolverParameter{fn}(int k, AcadosParameters& params, const double value, int index){{\n"
                        f"\t(void)index;\n\tparams.all_parameters[k * {npar} + {indices[0]}] = value;\n}}\n")
            else:
                h.write(f"void setSolverParameter{fn}(int k, AcadosParameters& params, const double value, int index);\n")
                c.write(f"void setSolverParameter{fn}(int k, AcadosParameters& params, const double value, int index){{\n")
                for i, idx in enumerate(indices):
                    c.write(("\tif" if i == 0 else "\telse if") + f"(index == {i})\n\t\tparams.all_parameters[k * {npar} + {idx}] = value;\n")
                c.write("}\n")
        h.write("}\n#endif\n"); c.write("}\n")
    return pm


if __name__ == "__main__":
    import sys
    generate_solver(sys.argv[1] if len(sys.argv) > 1 else "generated")
