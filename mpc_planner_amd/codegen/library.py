"""A library of cost / constraint modules written against plugin.py's protocol.  Each class states which reference
module it corresponds to (same parameter names and order, same expression), so that a configuration assembled from
these produces the reference's parameter map and stage functions -- tests/test_codegen.py checks both against the
golden vectors generated from the reference's own scripts."""
import numpy as np

from . import symbolic as cd
from .plugin import ConstraintModule, ObjectiveModule


def rotation_matrix(angle):
    """util/math.py:5-7."""
    return np.array([[cd.cos(angle), -cd.sin(angle)], [cd.sin(angle), cd.cos(angle)]])


# ---- objectives ---------------------------------------------------------------------------------------------------
class _Weights:
    def __init__(self):
        self._terms = []            # (variable, [weight names], function)

    def define_parameters(self, params):
        for _, names, _ in self._terms:
            for n in names:
                params.add(n, add_to_rqt_reconfigure=True)

    def get_value(self, model, params, settings, stage_idx):
        cost = 0.0
        for var, names, fn in self._terms:
            cost += fn(model.get(var), [params.get(n) for n in names])
        return cost


class MPCBaseModule(ObjectiveModule):
    """mpc_base.py: weigh_variable(var_name, weight_names, cost_function = w[0] * x**2)."""

    def __init__(self, settings=None):
        super().__init__()
        self.module_name, self.import_name = "MPCBaseModule", "mpc_base.h"
        self.objectives.append(_Weights())

    def add_definitions(self, header_file):
        """mpc_base.py:83-92: the weight names the C++ module reads from CONFIG["weights"]."""
        names = [n for _, ns, _ in self.objectives[0]._terms for n in ns]
        header_file.write("#define WEIGHT_PARAMS {" + ", ".join(f'"{n}"' for n in names) + "}\n")

    def weigh_variable(self, var_name, weight_names, cost_function=lambda x, w: w[0] * x ** 2, **_):
        names = [weight_names] if isinstance(weight_names, str) else list(weight_names)
        self.objectives[0]._terms.append((var_name, names, cost_function))


class _Segment:
    def __init__(self, params, name, i):
        self.a, self.b, self.c, self.d = (params.get(f"{name}{i}_{k}") for k in "abcd")
        self.start = params.get(f"spline{i}_start")

    def at(self, s):
        t = s - self.start
        return self.a * t * t * t + self.b * t * t + self.c * t + self.d

    def deriv(self, s):
        t = s - self.start
        return 3 * self.a * t * t + 2 * self.b * t + self.c


class GluedSpline:
    """spline.py:25-50: cubic segments glued with sigmoids 1/(1+exp((s - start_i + 0.02)/0.1)); values and segment
    derivatives are blended with the same weights."""

    def __init__(self, params, name, num_segments, s):
        self.segments = [_Segment(params, name, i) for i in range(num_segments)]
        self.lambdas = [1.0 / (1.0 + cd.exp((s - seg.start + 0.02) / 0.1)) for seg in self.segments[1:]]

    def _blend(self, values):
        value = values[-1]
        for k in range(len(values) - 1, 0, -1):
            value = self.lambdas[k - 1] * values[k - 1] + (1.0 - self.lambdas[k - 1]) * value
        return value

    def at(self, s):
        return self._blend([seg.at(s) for seg in self.segments])

    def deriv(self, s):
        return self._blend([seg.deriv(s) for seg in self.segments])


class _Contouring:
    def __init__(self, num_segments, dynamic_velocity_reference=False):
        self.num_segments = num_segments
        self.dynamic_velocity_reference = dynamic_velocity_reference

    def define_parameters(self, params):
        params.add("contour", add_to_rqt_reconfigure=True)
        params.add("lag", add_to_rqt_reconfigure=True)
        if not params.has_parameter("velocity"):        # contouring.py:26-28: only when no MPC base module defined them
            params.add("velocity", add_to_rqt_reconfigure=True)
            params.add("reference_velocity", add_to_rqt_reconfigure=True)
        params.add("terminal_angle", add_to_rqt_reconfigure=True)
        params.add("terminal_contouring", add_to_rqt_reconfigure=True)
        for i in range(self.num_segments):
            for ax in ("x", "y"):
                for k in "abcd":
                    params.add(f"spline_{ax}{i}_{k}", bundle_name=f"spline_{ax}_{k}")
            params.add(f"spline{i}_start", bundle_name="spline_start")

    def get_value(self, model, params, settings, stage_idx):
        x, y, s = model.get("x"), model.get("y"), model.get("spline")
        px = GluedSpline(params, "spline_x", self.num_segments, s)
        py = GluedSpline(params, "spline_y", self.num_segments, s)
        dx, dy = px.deriv(s), py.deriv(s)
        norm = cd.sqrt(dx * dx + dy * dy)
        tx, ty = dx / norm, dy / norm
        ex, ey = x - px.at(s), y - py.at(s)
        contour_error = ty * ex - tx * ey
        lag_error = tx * ex + ty * ey
        cost = params.get("lag") * lag_error ** 2 + params.get("contour") * contour_error ** 2
        if self.dynamic_velocity_reference:             # contouring.py:60-66,80-81: velocity reference from the path spline
            if not params.has_parameter("spline_v0_a"):
                raise IOError("contouring/dynamic_velocity_reference is enabled, but there is no PathReferenceVelocity module.")
            v_ref = GluedSpline(params, "spline_v", self.num_segments, s).at(s)
            cost += params.get("velocity") * (model.get("v") - v_ref) ** 2
        return cost


class ContouringModule(ObjectiveModule):
    """contouring.py:22-98 (stage cost; the terminal terms are Forces-only: the acados stage cost is built at stage_idx = 1)."""

    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "Contouring", "contouring.h"
        self.objectives.append(_Contouring(settings["contouring"]["num_segments"],
                                           settings["contouring"].get("dynamic_velocity_reference", False)))


class _PathReferenceVelocity:
    """Declares the velocity spline's parameters; the cost term itself lives in the contouring objective."""

    def __init__(self, num_segments):
        self.num_segments = num_segments

    def define_parameters(self, params):
        for i in range(self.num_segments):
            for k in "abcd":
                params.add(f"spline_v{i}_{k}", bundle_name=f"spline_v_{k}")

    def get_value(self, model, params, settings, stage_idx):
        return 0.0


class PathReferenceVelocityModule(ObjectiveModule):
    """path_reference_velocity.py:11-48."""

    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "PathReferenceVelocity", "path_reference_velocity.h"
        self.objectives.append(_PathReferenceVelocity(settings["contouring"]["num_segments"]))


class _Goal:
    def define_parameters(self, params):
        params.add("goal_weight", add_to_rqt_reconfigure=True)
        params.add("goal_x"); params.add("goal_y")

    def get_value(self, model, params, settings, stage_idx):
        gx, gy = params.get("goal_x"), params.get("goal_y")
        return params.get("goal_weight") * ((model.get("x") - gx) ** 2 + (model.get("y") - gy) ** 2) / (gx ** 2 + gy ** 2 + 0.01)


class GoalModule(ObjectiveModule):
    """goal_module.py:14-35."""

    def __init__(self, settings=None):
        super().__init__()
        self.module_name, self.import_name = "GoalModule", "goal_module.h"
        self.objectives.append(_Goal())


# ---- constraints --------------------------------------------------------------------------------------------------
def _disc_position(model, params, disc_id):
    pos = np.array([model.get("x"), model.get("y")])
    return pos + rotation_matrix(model.get("psi")).dot(np.array([params.get(f"ego_disc_{disc_id}_offset"), 0]))


class _Ellipsoids:
    def __init__(self, n_discs, max_obstacles):
        self.n_discs, self.max_obstacles = n_discs, max_obstacles
        self.nh = n_discs * max_obstacles

    def define_parameters(self, params):
        params.add("ego_disc_radius")
        for d in range(self.n_discs):
            params.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
        for j in range(self.max_obstacles):
            for f in ("x", "y", "psi", "major", "minor", "chi", "r"):
                params.add(f"ellipsoid_obst_{j}_{f}", bundle_name=f"ellipsoid_obst_{f}")

    def get_lower_bound(self):
        return [1.0] * self.nh

    def get_upper_bound(self):
        return [np.inf] * self.nh

    def get_constraints(self, model, params, settings, stage_idx):
        out = []
        r_disc = params.get("ego_disc_radius")
        for j in range(self.max_obstacles):
            g = lambda f: params.get(f"ellipsoid_obst_{j}_{f}")
            chi = g("chi")
            major, minor = g("major") * cd.sqrt(chi), g("minor") * cd.sqrt(chi)
            R = rotation_matrix(g("psi"))
            ab = np.array([[1.0 / ((major + r_disc + g("r")) ** 2), 0], [0, 1.0 / ((minor + r_disc + g("r")) ** 2)]])
            Q = R.T.dot(ab).dot(R)
            for d in range(self.n_discs):
                diff = _disc_position(model, params, d) - np.array([g("x"), g("y")])
                out.append(diff.dot(Q).dot(diff))
        return out


class EllipsoidConstraintModule(ConstraintModule):
    """ellipsoid_constraints.py:28-119."""

    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "EllipsoidConstraints", "ellipsoid_constraints.h"
        self.constraints.append(_Ellipsoids(settings["n_discs"], settings["max_obstacles"]))


class _Halfspaces:
    """a1 x + a2 y - b <= 0 (guidance_constraints.py:67-110), or on the disc position with the slack state relaxing b
    (decomp_constraints.py:34-98, scenario_constraints.py:30-94)."""

    def __init__(self, names, disc=None, use_slack=False, bundle=None):
        self.names, self.disc, self.use_slack, self.bundle = names, disc, use_slack, bundle
        self.nh = len(names)

    def define_parameters(self, params):
        if self.disc is not None:
            params.add(f"ego_disc_{self.disc}_offset", bundle_name="ego_disc_offset")
        for n in self.names:
            for f in ("a1", "a2", "b"):
                params.add(f"{n}_{f}", bundle_name=(f"{self.bundle}_{f}" if self.bundle else None))

    def get_lower_bound(self):
        return [-np.inf] * self.nh

    def get_upper_bound(self):
        return [0.0] * self.nh

    def get_constraints(self, model, params, settings, stage_idx):
        pos = _disc_position(model, params, self.disc) if self.disc is not None else np.array([model.get("x"), model.get("y")])
        slack = 0.0
        if self.use_slack and "slack" in model.states:
            slack = model.get("slack")
        return [params.get(f"{n}_a1") * pos[0] + params.get(f"{n}_a2") * pos[1] - (params.get(f"{n}_b") + slack) for n in self.names]


class GuidanceConstraintModule(ConstraintModule):
    """guidance_constraints.py:18-63: topology halfspaces + the collision-avoidance submodule's constraints."""

    def __init__(self, settings, constraint_submodule=EllipsoidConstraintModule):
        super().__init__()
        self.module_name, self.import_name = "GuidanceConstraints", "guidance_constraints.h"
        self.dependencies.append("guidance_planner")
        n = settings["max_obstacles"] + settings["linearized_constraints"]["add_halfspaces"]
        self.constraints.append(_Halfspaces([f"lin_constraint_{i}" for i in range(n)], bundle="lin_constraint"))
        self.sources.append("linearized_constraints.h")
        self.constraint_submodule = constraint_submodule(settings)
        self.constraints += self.constraint_submodule.constraints
        self.sources.append(self.constraint_submodule.import_name)

    def add_definitions(self, header_file):
        """guidance_constraints.py:56-62: which C++ module solves the per-topology problems."""
        header_file.write(f"#include <mpc_planner_modules/{self.constraint_submodule.import_name}>\n")
        header_file.write(f"#define GUIDANCE_CONSTRAINTS_TYPE {self.constraint_submodule.module_name}\n")


class DecompConstraintModule(ConstraintModule):
    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "DecompConstraints", "decomp_constraints.h"
        self.dependencies.append("decomp_util")
        n = settings["decomp"]["max_constraints"]
        self.constraints.append(_Halfspaces([f"disc_0_decomp_{i}" for i in range(n)], disc=0, use_slack=True, bundle="decomp"))


class ScenarioConstraintModule(ConstraintModule):
    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "ScenarioConstraints", "scenario_constraints.h"
        self.dependencies.append("scenario_module")
        self.constraints.append(_Halfspaces([f"disc_0_scenario_constraint_{i}" for i in range(24)], disc=0, use_slack=True))


class _GaussianChance:
    def __init__(self, n_discs, max_obstacles):
        self.n_discs, self.max_obstacles = n_discs, max_obstacles
        self.nh = n_discs * max_obstacles

    def define_parameters(self, params):
        params.add("ego_disc_radius")
        for d in range(self.n_discs):
            params.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
        for j in range(self.max_obstacles):
            for f in ("x", "y", "major", "minor", "risk", "r"):
                params.add(f"gaussian_obst_{j}_{f}", bundle_name=f"gaussian_obst_{f}")

    def get_lower_bound(self):
        return [0.0] * self.nh

    def get_upper_bound(self):
        return [np.inf] * self.nh

    def get_constraints(self, model, params, settings, stage_idx):
        out = []
        for j in range(self.max_obstacles):
            g = lambda f: params.get(f"gaussian_obst_{j}_{f}")
            sig2 = np.array([g("major") ** 2, g("minor") ** 2])
            xe = 1.0 - 2.0 * g("risk")
            # inverse error function: rational start + two Newton steps (gaussian_constraints.py:103-111)
            z = cd.sqrt(-cd.log((1.0 - xe) / 2.0))
            ye = (((1.641345311 * z + 3.429567803) * z - 1.624906493) * z - 1.970840454) / ((1.637067800 * z + 3.543889200) * z + 1.0)
            for _ in range(2):
                ye = ye - (cd.erf(ye) - xe) / (2.0 / cd.sqrt(cd.pi) * cd.exp(-ye * ye))
            for d in range(self.n_discs):
                diff = _disc_position(model, params, d) - np.array([g("x"), g("y")])
                a = diff / cd.sqrt(diff.dot(diff))
                out.append(a.dot(diff) - (params.get("ego_disc_radius") + g("r")) - ye * cd.sqrt(2.0 * (a * a).dot(sig2)))
        return out


class GaussianConstraintModule(ConstraintModule):
    """gaussian_constraints.py:14-113 (CC-MPC: linearised chance constraint through the Gaussian CDF)."""

    def __init__(self, settings):
        super().__init__()
        self.module_name, self.import_name = "GaussianConstraints", "gaussian_constraints.h"
        self.constraints.append(_GaussianChance(settings["n_discs"], settings["max_obstacles"]))
