"""Parameter map of one stage's parameter row.

Mirrors `Parameters.add/length/get` (solver_generator/util/parameters.py:10-96) and the ordering rule of
`define_parameters` (solver_generator/solver_definition.py:5-16): objective modules first, then constraint
modules, each in insertion order, duplicates skipped.  The per-module `define_parameters` bodies restated
here are: WeightsObjective (mpc_planner_modules/scripts/mpc_base.py:26-30 with the weigh_variable calls of
mpc_planner_jackalsimulator/scripts/generate_jackalsimulator_solver.py:45-52), ContouringObjective
(contouring.py:22-46), LinearConstraints (guidance_constraints.py:73-78), EllipsoidConstraint
(ellipsoid_constraints.py:37-49), and for the slack-model configurations (generate_jackalsimulator_solver.py:67-90,
generate_rosnavigation_solver.py:62-108) the scenario / decomp LinearConstraints (scenario_constraints.py:40-49,
decomp_constraints.py:44-52).
"""
from collections import OrderedDict


class ParameterMap:
    def __init__(self):
        self._params = OrderedDict()
        self.parameter_bundles = OrderedDict()

    def add(self, name, bundle_name=None):
        if name in self._params:
            return
        idx = len(self._params)
        self._params[name] = idx
        self.parameter_bundles.setdefault(bundle_name or name, []).append(idx)

    def length(self):
        return len(self._params)

    def has_parameter(self, name):
        return name in self._params

    def index(self, name):
        return self._params[name]

    def bundle(self, bundle_name):
        return self.parameter_bundles[bundle_name]

    def as_dict(self):
        d = dict(self._params)
        d["num parameters"] = self.length()      # save_map(), util/parameters.py:70-76
        return d


def define_parameters(num_segments, max_obstacles, guidance=True, add_halfspaces=0, n_discs=1, slack=False,
                      ellipsoids=True, n_scenario=0, n_decomp=0, gaussian=False):
    """T-MPC (guidance=True, cfg 2/4) or LMPCC-style basic (guidance=False, cfg 1) Jackal parameter map; with
    slack=True the slack-model maps: rosnavigation T-MPC (guidance + ellipsoids + n_decomp=12, cfg 3) and SH-MPC
    (guidance=False, ellipsoids=False, n_scenario=24, cfg 5)."""
    p = ParameterMap()
    # objective modules
    base = ("acceleration", "angular_velocity", "slack", "velocity", "reference_velocity") if slack else \
           ("acceleration", "angular_velocity", "velocity", "reference_velocity")
    for w in base:                                                                      # MPCBaseModule
        p.add(w)
    p.add("contour"); p.add("lag")                                                      # ContouringObjective
    p.add("terminal_angle"); p.add("terminal_contouring")
    for i in range(num_segments):
        for ax in ("x", "y"):
            for c in "abcd":
                p.add(f"spline_{ax}{i}_{c}", bundle_name=f"spline_{ax}_{c}")
        p.add(f"spline{i}_start", bundle_name="spline_start")
    # constraint modules
    if guidance:
        for j in range(max_obstacles + add_halfspaces):                                 # LinearConstraints
            p.add(f"lin_constraint_{j}_a1", bundle_name="lin_constraint_a1")
            p.add(f"lin_constraint_{j}_a2", bundle_name="lin_constraint_a2")
            p.add(f"lin_constraint_{j}_b", bundle_name="lin_constraint_b")
    if ellipsoids:
        p.add("ego_disc_radius")                                                        # EllipsoidConstraint
        for d in range(n_discs):
            p.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
        for j in range(max_obstacles):
            for f in ("x", "y", "psi", "major", "minor", "chi", "r"):
                p.add(f"ellipsoid_obst_{j}_{f}", bundle_name=f"ellipsoid_obst_{f}")
    if gaussian:                                                                        # GaussianConstraint (gaussian_constraints.py:40-52)
        p.add("ego_disc_radius")
        for d in range(n_discs):
            p.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
        for j in range(max_obstacles):
            for f in ("x", "y", "major", "minor", "risk", "r"):
                p.add(f"gaussian_obst_{j}_{f}", bundle_name=f"gaussian_obst_{f}")
    if n_scenario:                                                                      # scenario LinearConstraints
        for d in range(n_discs):
            p.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
            for j in range(n_scenario):
                for f in ("a1", "a2", "b"):
                    p.add(f"disc_{d}_scenario_constraint_{j}_{f}")
    if n_decomp:                                                                        # decomp LinearConstraints
        for d in range(n_discs):
            p.add(f"ego_disc_{d}_offset", bundle_name="ego_disc_offset")
            for j in range(n_decomp):
                for f in ("a1", "a2", "b"):
                    p.add(f"disc_{d}_decomp_{j}_{f}", bundle_name=f"decomp_{f}")
    return p


# model_map.yaml content for ContouringSecondOrderUnicycleModel (solver_model.py:118-128, 193-205):
# name -> ["x"|"u", index in z=[u;x], lower bound, upper bound]
import math

MODEL_MAP_UNICYCLE = OrderedDict([
    ("a", ["u", 0, -2.0, 2.0]), ("w", ["u", 1, -0.8, 0.8]),
    ("x", ["x", 2, -2000.0, 2000.0]), ("y", ["x", 3, -2000.0, 2000.0]),
    ("psi", ["x", 4, -math.pi * 4, math.pi * 4]), ("v", ["x", 5, -0.01, 3.0]),
    ("spline", ["x", 6, -1.0, 10000.0]),
])
# ContouringSecondOrderUnicycleModelWithSlack (solver_model.py:274-298)
MODEL_MAP_UNICYCLE_SLACK = OrderedDict(list(MODEL_MAP_UNICYCLE.items()) + [("slack", ["x", 7, 0.0, 5000.0])])
