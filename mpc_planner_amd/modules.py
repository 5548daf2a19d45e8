"""Host-side mirror of the C++ modules' parameter writers on the T-MPC path (numpy, per trajectory).

Each function cites the reference method it restates; they fill one solver's `all_parameters[N][npar]`
(acados_solver_interface.h:56) and warm start `x0[(N+1)][nvar]` (:54) exactly like the reference's
`setParameters(k)` / `initializeSolverWithGuidance` do, so the parity tests read like the reference flow.
"""
import numpy as np

NU, NX, NV = 2, 5, 7
IDX = dict(a=0, w=1, x=2, y=3, psi=4, v=5, spline=6)   # model_map.yaml order (solver_model.py:118-128)


def mpc_base_set_parameters(pm, params, weights):
    """MPCBaseModule::setParameters (mpc_planner_modules/src/mpc_base.cpp:23-35): every stage k."""
    for name in ("acceleration", "angular_velocity", "velocity", "reference_velocity"):
        params[:, pm.index(name)] = weights[name]


def contouring_set_parameters(pm, params, weights, segments):
    """Contouring::setParameters + setSplineParameters (contouring.cpp:50-124): weights and the S
    segments starting at the closest one, identical for every stage k.
    segments: array [S][9] = (xa,xb,xc,xd, ya,yb,yc,yd, start)."""
    for name in ("contour", "lag", "terminal_angle", "terminal_contouring"):
        params[:, pm.index(name)] = weights[name]
    names = ["spline_x{}_a", "spline_x{}_b", "spline_x{}_c", "spline_x{}_d",
             "spline_y{}_a", "spline_y{}_b", "spline_y{}_c", "spline_y{}_d", "spline{}_start"]
    for i in range(segments.shape[0]):
        for w, n in enumerate(names):
            params[:, pm.index(n.format(i))] = segments[i, w]


def ellipsoid_set_parameters(pm, params, state_xy, obstacles, robot_radius, disc_offset=0.0):
    """EllipsoidConstraints::update/setParameters (ellipsoid_constraints.cpp:24-90).
    obstacles: dict(pos [M][N][2], angle [M][N], radius [M], major [M][N], minor [M][N], chi [M])."""
    N = params.shape[0]
    M = obstacles["pos"].shape[0]
    params[:, pm.index("ego_disc_radius")] = robot_radius
    params[:, pm.index("ego_disc_0_offset")] = disc_offset
    for j in range(M):
        ix = [pm.index(f"ellipsoid_obst_{j}_{f}") for f in ("x", "y", "psi", "major", "minor", "chi", "r")]
        # k == 0: dummies (:42-56)
        params[0, ix] = [state_xy[0] + 50.0, state_xy[1] + 50.0, 0.0, 0.0, 0.0, 1.0, 0.1]
        # k >= 1: prediction step k-1 (:62-85)
        params[1:, ix[0]] = obstacles["pos"][j, :N - 1, 0]
        params[1:, ix[1]] = obstacles["pos"][j, :N - 1, 1]
        params[1:, ix[2]] = obstacles["angle"][j, :N - 1]
        params[1:, ix[3]] = obstacles["major"][j, :N - 1]
        params[1:, ix[4]] = obstacles["minor"][j, :N - 1]
        params[1:, ix[5]] = obstacles["chi"][j]
        params[1:, ix[6]] = obstacles["radius"][j]


def linearized_update(x0, obstacle_pos, robot_radius):
    """LinearizedConstraints::update in guidance mode (linearized_constraints.cpp:49-105).
    x0: warm start [N+1][nvar]; obstacle_pos [M][N][2].  Returns a1,a2,b [N][M] (row k=0 unused).
    The Douglas-Rachford projection (projectToSafety :130-148, ros_tools source absent) is the identity
    whenever the guess is already >= r+robot_radius away from every obstacle; synthetic scenes guarantee
    that, and this mirror asserts it."""
    Np1 = x0.shape[0]; N = Np1 - 1; M = obstacle_pos.shape[0]
    a1 = np.zeros((N, M)); a2 = np.zeros((N, M)); b = np.zeros((N, M))
    radius = 1e-3                                             # _use_guidance (:99)
    for k in range(1, N):
        pos = x0[k, [IDX["x"], IDX["y"]]]
        for j in range(M):
            o = obstacle_pos[j, k - 1]
            diff = o - pos
            dist = np.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
            assert dist >= radius + robot_radius, "guess inside the projection radius: DR projection not restated"
            a1[k, j] = diff[0] / dist
            a2[k, j] = diff[1] / dist
            b[k, j] = a1[k, j] * o[0] + a2[k, j] * o[1] - (radius + robot_radius)
    return a1, a2, b


def linearized_set_parameters(pm, params, state_x, lin=None, n_rows=None):
    """LinearizedConstraints::setParameters (linearized_constraints.cpp:150-189).  lin=None writes the
    all-dummy rows of the non-guided T-MPC++ planner (guidance_constraints.cpp:301-305,323-324)."""
    N = params.shape[0]
    dummy = (1.0, 0.0, state_x + 100.0)                       # _dummy_a1,_dummy_a2 (header), _dummy_b (:54)
    for j in range(n_rows):
        ia = [pm.index(f"lin_constraint_{j}_{f}") for f in ("a1", "a2", "b")]
        params[:, ia] = dummy                                 # k == 0 and unused rows
        if lin is not None and j < lin[0].shape[1]:
            params[1:, ia[0]] = lin[0][1:, j]
            params[1:, ia[1]] = lin[1][1:, j]
            params[1:, ia[2]] = lin[2][1:, j]


def initialize_with_forward_propagation(state, N, dt):
    """Main-solver warm start: constant-velocity forward propagation of the current state (stand-in for
    the previous tick's solution used by initializeWarmstart, acados_solver_interface.cpp:344-376; same
    recursion as initializeWithBraking :303-342 with a = 0)."""
    x0 = np.zeros((N + 1, NV))
    x, y, psi, v, s = state
    for k in range(N + 1):
        x0[k] = [0.0, 0.0, x, y, psi, v, s]
        x += v * dt * np.cos(psi); y += v * dt * np.sin(psi); s += v * dt
    return x0


def initialize_solver_with_guidance(x0, guidance_pos, guidance_vel):
    """GuidanceConstraints::initializeSolverWithGuidance (guidance_constraints.cpp:390-414):
    for k = 1..N-1 set x,y from the guidance spline at t = k*dt, psi = atan2(vy,vx), v = |vel|."""
    N = x0.shape[0] - 1
    for k in range(1, N):
        x0[k, IDX["x"]] = guidance_pos[k, 0]
        x0[k, IDX["y"]] = guidance_pos[k, 1]
        x0[k, IDX["psi"]] = np.arctan2(guidance_vel[k, 1], guidance_vel[k, 0])
        x0[k, IDX["v"]] = np.sqrt(guidance_vel[k, 0] ** 2 + guidance_vel[k, 1] ** 2)
    return x0
