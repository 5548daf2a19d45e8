"""The C++ wiring files the reference's generator writes for its module classes next to a solver
(solver_generator/generate_cpp_files.py:11-95 -- consumed by mpc_planner_modules' own C++ sources, which a drop-in
`Solver` keeps unchanged):

  include/mpc_planner_modules/modules.h       includes + `initializeModules(modules, solver)` factory, one
                                              `std::make_shared<ModuleName>(solver)` per module in stack order
  include/mpc_planner_modules/definitions.h   whatever each module's `add_definitions(file)` writes (WEIGHT_PARAMS,
                                              GUIDANCE_CONSTRAINTS_TYPE, ...)
  modules.cmake                               MODULE_DEPENDENCIES / MODULE_SOURCES lists

Works on anything following the module protocol (plugin.py): `.module_name`, `.import_name`, `.sources`, `.dependencies`,
`.add_definitions(file)` -- the reference's own module objects included.  Modules without an `import_name` (pure python
cost terms with no C++ counterpart) are skipped in modules.h / modules.cmake.
"""
import io
import os


def _stem(header):
    return header.split(".")[0]


def _unique(items):
    seen = []
    for x in items:
        if x not in seen:
            seen.append(x)
    return seen


def modules_header(modules):
    cpp = [m for m in modules.modules if getattr(m, "import_name", None)]
    out = io.StringIO()
    out.write("#ifndef __MPC_PLANNER_GENERATED_MODULES_H__\n#define __MPC_PLANNER_GENERATED_MODULES_H__\n\n")
    for m in cpp:
        out.write(f"#include <mpc_planner_modules/{m.import_name}>\n")
        for s in getattr(m, "sources", []):
            out.write(f"#include <mpc_planner_modules/{_stem(s)}.h>\n")
    out.write("\nnamespace MPCPlanner\n{\n\tclass Solver;\n"
              "\tinline void initializeModules(std::vector<std::shared_ptr<ControllerModule>> &modules, "
              "std::shared_ptr<Solver> solver)\n\t{\n")
    for m in cpp:
        out.write(f"\t\tmodules.emplace_back(nullptr);\n\t\tmodules.back() = std::make_shared<{m.module_name}>(solver);\n")
    out.write("\n\t}\n}\n#endif")
    return out.getvalue()


def definitions_header(modules):
    out = io.StringIO()
    out.write("#ifndef __MPC_PLANNER_GENERATED_DEFINITIONS_H__\n#define __MPC_PLANNER_GENERATED_DEFINITIONS_H__\n\n")
    for m in modules.modules:
        if hasattr(m, "add_definitions"):
            m.add_definitions(out)
    out.write("\n\n#endif")
    return out.getvalue()


def modules_cmake(modules):
    cpp = [m for m in modules.modules if getattr(m, "import_name", None)]
    deps = _unique(d for m in cpp for d in getattr(m, "dependencies", []))
    out = io.StringIO()
    out.write("if(USE_ROS2)\n" + "".join(f"\tfind_package({d} REQUIRED)\n" for d in deps) + "endif()\n")
    out.write("set(MODULE_DEPENDENCIES\n" + "".join(f"\t{d}\n" for d in deps) + ")\n\n")
    out.write("set(MODULE_SOURCES\n")
    for m in cpp:
        out.write(f"\tsrc/{_stem(m.import_name)}.cpp\n")
        for s in getattr(m, "sources", []):
            out.write(f"\tsrc/{_stem(s)}.cpp\n")
    out.write(")\n")
    return out.getvalue()


def write_module_glue(out_dir, modules):
    inc = os.path.join(out_dir, "include", "mpc_planner_modules")
    os.makedirs(inc, exist_ok=True)
    files = {os.path.join(inc, "modules.h"): modules_header(modules),
             os.path.join(inc, "definitions.h"): definitions_header(modules),
             os.path.join(out_dir, "modules.cmake"): modules_cmake(modules)}
    for path, text in files.items():
        with open(path, "w") as fh:
            fh.write(text)
    return sorted(files)
