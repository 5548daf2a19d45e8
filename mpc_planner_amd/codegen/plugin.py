"""The module protocol the generator consumes -- same surface as the reference's solver_generator, so that module
scripts written for it plug in here:

  Parameters            add / get / has_parameter / load / length            (util/parameters.py:10-96)
  ModuleManager         add_module, .modules                                 (control_modules.py:4-36)
  ObjectiveModule       .type == "objective", .objectives[*].define_parameters(params) / .get_value(model, params, settings, stage_idx)
  ConstraintModule      .type == "constraint", .constraints[*].define_parameters / .get_constraints / .get_lower_bound /
                        .get_upper_bound / .nh                               (control_modules.py:39-113)
  model                 .states, .inputs, .nu, .nx, .lower_bound, .upper_bound, .load(z), .get(name)  (solver_model.py:54-168)
  define_parameters / objective / constraints / bounds                       (solver_definition.py:5-76)

Anything that quacks like these (e.g. the reference's own classes, imported by the user) is accepted by emit.py.
"""
import math

from ..parameters import ParameterMap


class Parameters(ParameterMap):
    """ParameterMap + the symbolic side (`load(p)`, `get(name)`) the module scripts use while building expressions."""

    def __init__(self):
        super().__init__()
        self._p = None

    def add(self, parameter, add_to_rqt_reconfigure=False, rqt_config_name=None, bundle_name=None, rqt_min_value=0.0,
            rqt_max_value=100.0):
        super().add(parameter, bundle_name=bundle_name)

    def load(self, p):
        self._p = p

    def get(self, parameter):
        if self._p is None:
            raise RuntimeError("Parameters.load(p) was not called")
        return self._p[self._params[parameter]]

    def get_p(self):
        return self._p


class ModuleManager:
    def __init__(self):
        self.modules = []

    def add_module(self, module):
        self.modules.append(module)
        return module


class Module:
    """control_modules.py:39-56.  module_name / import_name name the C++ class and header of the module in
    mpc_planner_modules (cpp_glue.py writes modules.h / definitions.h / modules.cmake from them)."""

    def __init__(self):
        self.module_name = "UNDEFINED"
        self.import_name = None
        self.description = ""
        self.submodules = []
        self.dependencies = []
        self.sources = []

    def add_definitions(self, header_file):
        pass


class ObjectiveModule(Module):
    def __init__(self):
        super().__init__()
        self.type = "objective"
        self.objectives = []

    def define_parameters(self, params):
        for o in self.objectives:
            o.define_parameters(params)

    def get_value(self, model, params, settings, stage_idx):
        cost = 0.0
        for o in self.objectives:
            cost += o.get_value(model, params, settings, stage_idx)
        return cost


class ConstraintModule(Module):
    def __init__(self):
        super().__init__()
        self.type = "constraint"
        self.constraints = []

    def define_parameters(self, params):
        for c in self.constraints:
            c.define_parameters(params)


# ---- models (variable naming / bounds only: the kernels integrate the unicycle in closed form) -------------------
class UnicycleContouringModel:
    """ContouringSecondOrderUnicycleModel (solver_model.py:193-214)."""

    def __init__(self):
        self.nu, self.nx = 2, 5
        self.inputs = ["a", "w"]
        self.states = ["x", "y", "psi", "v", "spline"]
        self.lower_bound = [-2.0, -0.8, -2000.0, -2000.0, -math.pi * 4, -0.01, -1.0]
        self.upper_bound = [2.0, 0.8, 2000.0, 2000.0, math.pi * 4, 3.0, 10000.0]
        self._z = None

    def get_nvar(self):
        return self.nu + self.nx

    def load(self, z):
        self._z = z

    def get(self, name):
        if name in self.states:
            return self._z[self.nu + self.states.index(name)]
        if name in self.inputs:
            return self._z[self.inputs.index(name)]
        raise IOError(f"Requested a state or input `{name}' that was neither a state nor an input for the selected model")

    def get_x(self):
        return self._z[self.nu:]

    def get_u(self):
        return self._z[:self.nu]

    def get_xinit(self):
        return range(self.nu, self.get_nvar())

    def _index(self, name):
        if name in self.states:
            return self.nu + self.states.index(name)
        if name in self.inputs:
            return self.inputs.index(name)
        raise IOError(f"Requested a state or input `{name}' that was neither a state nor an input for the selected model")

    def set_bounds(self, lower_bound, upper_bound):
        if not len(lower_bound) == len(upper_bound) == len(self.lower_bound):
            raise ValueError("bounds must cover every input and state")
        self.lower_bound, self.upper_bound = list(lower_bound), list(upper_bound)

    def get_bounds(self, name):
        """(lower, upper, range) of an input or state (solver_model.py:151-168)."""
        i = self._index(name)
        return self.lower_bound[i], self.upper_bound[i], self.upper_bound[i] - self.lower_bound[i]

    def continuous_model(self, x, u):
        """xdot = [v cos psi, v sin psi, w, a, v] (solver_model.py:207-214) -- for inspection and tests: the kernels integrate
        exactly this model in closed form (csrc/tmpc_stage.hpp, ERK4 x 3 per stage)."""
        a, w = u[0], u[1]
        psi, v = x[2], x[3]
        return [v * math.cos(psi), v * math.sin(psi), w, a, v]


class SecondOrderUnicycleModel(UnicycleContouringModel):
    """SecondOrderUnicycleModel (solver_model.py:170-191): states x, y, psi, v -- no spline state; the stacks that track a goal instead of a
    reference path use it.  The kernels run it on their 5-state layout with the fifth slot inert (s' = 0): emit.py marks the generated
    library (tmpc_gen::MODEL = 1), callers pad xinit / x0 / read xtraj with stride 5 / 7 and a zero in the last slot."""

    def __init__(self):
        super().__init__()
        self.nx = 4
        self.states = ["x", "y", "psi", "v"]
        self.lower_bound = [-2.0, -2.0, -200.0, -200.0, -math.pi * 4, -2.0]
        self.upper_bound = [2.0, 2.0, 200.0, 200.0, math.pi * 4, 3.0]

    def continuous_model(self, x, u):
        a, w = u[0], u[1]
        psi, v = x[2], x[3]
        return [v * math.cos(psi), v * math.sin(psi), w, a]


class UnicycleContouringSlackModel(UnicycleContouringModel):
    """ContouringSecondOrderUnicycleModelWithSlack (solver_model.py:274-298)."""

    def __init__(self):
        super().__init__()
        self.nx = 6
        self.states = self.states + ["slack"]
        self.lower_bound = self.lower_bound + [0.0]
        self.upper_bound = self.upper_bound + [5000.0]

    def continuous_model(self, x, u):
        return super().continuous_model(x, u) + [0.0]                      # slack' = 0 (solver_model.py:287-295)

    def get_xinit(self):
        return range(self.nu, self.get_nvar() - 1)                          # the slack is not an initial state (:297-298)


# ---- assembly (solver_definition.py:5-76) ------------------------------------------------------------------------
def define_parameters(modules, params, settings):
    for kind in ("objective", "constraint"):
        for m in modules.modules:
            if m.type == kind:
                m.define_parameters(params)
    return params


def objective(modules, z, p, model, settings, stage_idx):
    params = settings["params"]
    params.load(p); model.load(z)
    cost = 0.0
    for m in modules.modules:
        if m.type == "objective":
            cost += m.get_value(model, params, settings, stage_idx)
    return cost


def constraints(modules, z, p, model, settings, stage_idx):
    params = settings["params"]
    params.load(p); model.load(z)
    out = []
    for m in modules.modules:
        if m.type == "constraint":
            for c in m.constraints:
                out += list(c.get_constraints(model, params, settings, stage_idx))
    return out


def constraint_bounds(modules):
    lb, ub = [], []
    for m in modules.modules:
        if m.type == "constraint":
            for c in m.constraints:
                lb += list(c.get_lower_bound()); ub += list(c.get_upper_bound())
    return lb, ub
