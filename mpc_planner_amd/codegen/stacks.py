"""Module stacks of the reference's shipped configurations, assembled from library.py (for tests, examples and
`__graft_entry__.build()`'s demo library)."""
from . import library as L
from . import plugin as P


def settings(N=20, max_obstacles=8, num_segments=5):
    return {"N": N, "n_discs": 1, "max_obstacles": max_obstacles, "linearized_constraints": {"add_halfspaces": 0},
            "contouring": {"num_segments": num_segments, "dynamic_velocity_reference": False},
            "decomp": {"range": 2.0, "max_constraints": 12}}


def _base(st, slack=False):
    mm = P.ModuleManager()
    b = mm.add_module(L.MPCBaseModule(st))
    b.weigh_variable("a", "acceleration"); b.weigh_variable("w", "angular_velocity")
    if slack:
        b.weigh_variable("slack", "slack")
    b.weigh_variable("v", ["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    mm.add_module(L.ContouringModule(st))
    return mm


def tmpc(st):
    """generate_jackalsimulator_solver.py:92-101 configuration_tmpc (BASELINE cfg 2 / 4)."""
    mm = _base(st); mm.add_module(L.GuidanceConstraintModule(st))
    return P.UnicycleContouringModel(), mm


def basic(st):
    """generate_jackalsimulator_solver.py:59-64 configuration_basic (BASELINE cfg 1)."""
    mm = _base(st); mm.add_module(L.EllipsoidConstraintModule(st))
    return P.UnicycleContouringModel(), mm


def safe_horizon(st):
    """generate_jackalsimulator_solver.py:67-90 configuration_safe_horizon (BASELINE cfg 5)."""
    mm = _base(st, slack=True); mm.add_module(L.ScenarioConstraintModule(st))
    return P.UnicycleContouringSlackModel(), mm


def rosnav_tmpc(st):
    """generate_rosnavigation_solver.py:86-108 configuration_tmpc (BASELINE cfg 3)."""
    mm = _base(st, slack=True); mm.add_module(L.GuidanceConstraintModule(st)); mm.add_module(L.DecompConstraintModule(st))
    return P.UnicycleContouringSlackModel(), mm


def jackal_tmpc(st):
    """generate_jackal_solver.py:53-73 configuration_tmpc -- mpc_planner_jackal's default: T-MPC whose collision-avoidance
    submodule is the Gaussian chance constraint."""
    mm = _base(st); mm.add_module(L.GuidanceConstraintModule(st, constraint_submodule=L.GaussianConstraintModule))
    return P.UnicycleContouringModel(), mm


def goal_gaussian(st):
    """A stack the hand-written kernels do not cover: goal tracking + Gaussian chance constraints (CC-MPC)."""
    mm = P.ModuleManager()
    b = mm.add_module(L.MPCBaseModule(st))
    b.weigh_variable("a", "acceleration"); b.weigh_variable("w", "angular_velocity")
    b.weigh_variable("v", ["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    mm.add_module(L.GoalModule(st)); mm.add_module(L.GaussianConstraintModule(st))
    return P.UnicycleContouringModel(), mm


def goal_ellipsoids_second_order_unicycle(st):
    """Goal tracking with ellipsoidal collision avoidance on SecondOrderUnicycleModel (solver_model.py:170-191; goal_module.py:22-36,
    ellipsoid_constraints.py:66-110): a stack on the model WITHOUT a spline state -- the kernels' fifth state slot is inert."""
    mm = P.ModuleManager()
    b = mm.add_module(L.MPCBaseModule(st))
    b.weigh_variable("a", "acceleration"); b.weigh_variable("w", "angular_velocity")
    b.weigh_variable("v", ["velocity", "reference_velocity"], cost_function=lambda x, w: w[0] * (x - w[1]) ** 2)
    mm.add_module(L.GoalModule(st)); mm.add_module(L.EllipsoidConstraintModule(st))
    return P.SecondOrderUnicycleModel(), mm


def contouring_path_velocity_ellipsoids(st):
    """The stack of the reference's own generation test (solver_generator/test/test_acados.py:30-46): MPC base weights on
    a, w + contouring + path reference velocity + ellipsoids."""
    mm = P.ModuleManager()
    b = mm.add_module(L.MPCBaseModule(st))
    b.weigh_variable("a", "acceleration"); b.weigh_variable("w", "angular_velocity")
    mm.add_module(L.ContouringModule(st)); mm.add_module(L.PathReferenceVelocityModule(st))
    mm.add_module(L.EllipsoidConstraintModule(st))
    return P.UnicycleContouringModel(), mm
