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
    for name in ("acceleration", "angular_velocity", "slack", "velocity", "reference_velocity"):
        if pm.has_parameter(name):
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


def gaussian_set_parameters(pm, params, state_xy, obstacles, robot_radius, risk, disc_offset=0.0):
    """GaussianConstraints::update/setParameters (gaussian_constraints.cpp:22-79): stage 0 dummies (x + 100, y + 100,
    0.1, 0.1, 0.05, 0.1), stages k >= 1 the Gaussian prediction step k-1 with CONFIG probabilistic/risk and obstacle_radius."""
    N = params.shape[0]
    M = obstacles["pos"].shape[0]
    params[:, pm.index("ego_disc_radius")] = robot_radius
    params[:, pm.index("ego_disc_0_offset")] = disc_offset
    for j in range(M):
        ix = [pm.index(f"gaussian_obst_{j}_{f}") for f in ("x", "y", "major", "minor", "risk", "r")]
        params[0, ix] = [state_xy[0] + 100.0, state_xy[1] + 100.0, 0.1, 0.1, 0.05, 0.1]
        params[1:, ix[0]] = obstacles["pos"][j, :N - 1, 0]
        params[1:, ix[1]] = obstacles["pos"][j, :N - 1, 1]
        params[1:, ix[2]] = obstacles["major"][j, :N - 1]
        params[1:, ix[3]] = obstacles["minor"][j, :N - 1]
        params[1:, ix[4]] = risk
        params[1:, ix[5]] = obstacles["radius"][j]


def _outside_disc(qx, qy, cx, cy, r):
    """Nearest point of the closed outside of the disc (c, r)."""
    dx, dy = qx - cx, qy - cy
    dist = np.sqrt(dx * dx + dy * dy)
    if dist >= r:
        return qx, qy
    if dist > 1e-12:
        s = r / dist
        return cx + dx * s, cy + dy * s
    return cx, cy + r


def project_to_safety(pos, obstacles_k, r):
    """(r: one radius for every obstacle, or one per obstacle -- the `_use_guidance == false` branch uses each obstacle's own.)"""
    return _project_to_safety(pos, obstacles_k, np.broadcast_to(np.asarray(r, float), (len(obstacles_k),)))


def _project_to_safety(pos, obstacles_k, radii):
    """LinearizedConstraints::projectToSafety (linearized_constraints.cpp:130-148): at most 3 sweeps over the obstacles of
    ros_tools' Douglas-Rachford projection, with obstacle 0 as the anchor.  The ros_tools source is not in the reference tree
    (DESIGN.md U10); restated from the published Douglas-Rachford operator p <- (p + R_delta R_anchor p) / 2 with reflections
    R = 2 P - I, P = nearest point outside the disc of radius r, applied when p is inside the obstacle's disc -- the call order of
    the reference (anchor first, then the obstacle).  The identity for a guess clear of every obstacle.  Same arithmetic as
    tmpc_linearize_topology_kernel (csrc/tmpc_aux_kernels.hpp) and the C++ DouglasRachford (modules_hip.h)."""
    px, py = float(pos[0]), float(pos[1])
    if len(obstacles_k) == 0:
        return np.array([px, py])
    ax, ay = float(obstacles_k[0][0]), float(obstacles_k[0][1])
    for _ in range(3):
        for o, r in zip(obstacles_k, radii):
            dx, dy = px - o[0], py - o[1]
            if np.sqrt(dx * dx + dy * dy) < r:
                qx, qy = _outside_disc(px, py, ax, ay, r)
                rx, ry = 2.0 * qx - px, 2.0 * qy - py
                bx, by = _outside_disc(rx, ry, float(o[0]), float(o[1]), r)
                sx, sy = 2.0 * bx - rx, 2.0 * by - ry
                px, py = (px + sx) / 2.0, (py + sy) / 2.0
    return np.array([px, py])


def linearized_update(x0, obstacle_pos, robot_radius, obstacle_radius=None, static=None):
    """LinearizedConstraints::update (linearized_constraints.cpp:49-123).  x0: warm start [N+1][nvar]; obstacle_pos [M][N][2].
    obstacle_radius None: guidance mode, radius 1e-3 (:99); else [M], the `_use_guidance == false` branch with each obstacle's own radius.
    static: [N][n_static][3] static halfspaces (a1, a2, b) per stage, appended behind the obstacle rows as they are (:107-123).
    Returns a1, a2, b [N][M + n_static] (row k = 0 unused)."""
    Np1 = x0.shape[0]; N = Np1 - 1; M = obstacle_pos.shape[0]
    n_static = 0 if static is None else static.shape[1]
    a1 = np.zeros((N, M + n_static)); a2 = np.zeros((N, M + n_static)); b = np.zeros((N, M + n_static))
    radii = (np.full(M, 1e-3) if obstacle_radius is None else np.asarray(obstacle_radius, float)) + robot_radius   # _use_guidance (:99, :140)
    for k in range(1, N):
        pos = project_to_safety(x0[k, [IDX["x"], IDX["y"]]], obstacle_pos[:, k - 1], radii) if M else x0[k, [IDX["x"], IDX["y"]]]
        for j in range(M):
            o = obstacle_pos[j, k - 1]
            diff = o - pos
            dist = np.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
            a1[k, j] = diff[0] / dist
            a2[k, j] = diff[1] / dist
            b[k, j] = a1[k, j] * o[0] + a2[k, j] * o[1] - radii[j]
        if n_static:
            a1[k, M:] = static[k, :, 0]; a2[k, M:] = static[k, :, 1]; b[k, M:] = static[k, :, 2]
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


def halfspace_rows_set_parameters(pm, params, state_x, rows, prefix, n_rows, disc_offset=0.0):
    """DecompConstraints::setParameters (decomp_constraints.cpp:150-187) and the scenario_module's parameter writer
    (scenario_constraints.cpp:76-79; same layout, scenario_constraints.py:40-49): ego disc offset at every stage,
    k = 0 all dummies (1, 0, x + 100), stages k >= 1 the rows computed by update(), padded with dummies.
    rows: (a1, a2, b) each [N][n] with row k = 0 unused, or None; prefix: "disc_0_decomp" | "disc_0_scenario_constraint"."""
    params[:, pm.index("ego_disc_0_offset")] = disc_offset
    dummy = (1.0, 0.0, state_x + 100.0)
    for j in range(n_rows):
        ia = [pm.index(f"{prefix}_{j}_{f}") for f in ("a1", "a2", "b")]
        params[:, ia] = dummy
        if rows is not None and j < rows[0].shape[1]:
            ok = ~np.isnan(rows[0][1:, j])                   # NaN = no row computed for this slot: stays a dummy
            for w in range(3):
                col = params[1:, ia[w]]
                col[ok] = rows[w][1:, j][ok]
                params[1:, ia[w]] = col


POLY_EPS_PARALLEL = 1e-12       # |sin| below which two halfspace boundaries count as parallel
POLY_TOL_EDGE = 1e-9            # minimal length [m] of the piece of a boundary line that lies on the polygon
POLY_SEED_MARGIN = 1e-6         # slack of the candidate filter
POLY_BINS = (4, 32, 256)        # sector resolution (bins per octant) of the three filter rounds


def _poly_sector(ax, ay, bins):
    """Direction sector 0..8*bins-1 of unit vectors (octant x bins of min(|ax|,|ay|)/max(|ax|,|ay|)): only comparisons, one divide
    and one multiply by a power of two, so host and device classify identically."""
    u, v = np.abs(ax), np.abs(ay)
    steep = v > u
    octant = (ax < 0).astype(int) | ((ay < 0).astype(int) << 1) | (steep.astype(int) << 2)
    t = np.where(steep, u, v) / np.where(steep, v, u)
    return octant * bins + np.minimum((t * float(bins)).astype(int), bins - 1)


def _poly_clip(ax, ay, dm, rows, cols, kill_below, dedup):
    """Clip the boundary line of every halfspace in `rows` (q_i + t perp_i, q_i = p + dm_i a_i) with the halfspaces in `cols`:
    (a_j . perp_i) t <= dm_j - dm_i (a_j . a_i).  Returns lo, hi (the interval of t left) and kill (a parallel halfspace
    excludes the whole line -- kill_below <= 0 is the slack of that decision; with dedup, of identical halfspaces only the
    lowest index survives)."""
    a1i, a2i, di = ax[rows][:, None], ay[rows][:, None], dm[rows][:, None]
    a1j, a2j, dj = ax[cols][None, :], ay[cols][None, :], dm[cols][None, :]
    c = a1j * (-a2i) + a2j * a1i
    dot = a1j * a1i + a2j * a2i
    rhs = dj - di * dot
    other = np.asarray(rows)[:, None] != np.asarray(cols)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = rhs / c
    hi = np.where(other & (c > POLY_EPS_PARALLEL), ratio, np.inf).min(axis=1, initial=np.inf)
    lo = np.where(other & (c < -POLY_EPS_PARALLEL), ratio, -np.inf).max(axis=1, initial=-np.inf)
    par = other & (np.abs(c) <= POLY_EPS_PARALLEL)
    same = dot > 0.0                                   # same direction: the closer one wins; opposite: an empty strip kills both
    k = par & np.where(same, dj < di + kill_below, rhs < kill_below)
    if dedup:
        k |= par & same & (dj == di) & (np.asarray(cols)[None, :] < np.asarray(rows)[:, None])
    return lo, hi, k.any(axis=1)


def polygon_edges(ax, ay, dm):
    """Which of the halfspaces  a_i . (q - p) <= dm_i  (unit normals a_i = (ax, ay), margins dm_i measured from the point p they
    were linearised around) form the boundary of their intersection polygon: halfspace i is an edge iff the piece of its boundary
    line inside all other halfspaces has positive length (every other halfspace is redundant -- removing it changes nothing).
    A filter comes first -- free to be anything conservative, since a halfspace whose boundary line misses the polygon of SOME of
    the halfspaces (the seeds) cannot touch the smaller polygon of all, and dropping it changes neither the polygon nor its edges:
    here three rounds at sector resolution POLY_BINS, the closest halfspace of each direction sector a seed (lowest index on ties),
    every seed clipping; tmpc_scenario_halfspaces_kernel filters differently (neighbouring seeds only, no division).  Then the edge
    test above among the candidates, which the kernel runs with the same per-pair arithmetic -- the rows are equal bit for bit.
    Returns a bool array."""
    n = len(dm)
    cand = np.arange(n)
    for bins in POLY_BINS:
        sec = _poly_sector(ax[cand], ay[cand], bins)
        order = np.lexsort((cand, dm[cand]))                    # by margin, then index
        seeds = np.sort(cand[order[np.unique(sec[order], return_index=True)[1]]])
        lo, hi, kill = _poly_clip(ax, ay, dm, cand, seeds, -POLY_SEED_MARGIN, False)
        cand = cand[~kill & (hi - lo > -POLY_SEED_MARGIN)]
    lo, hi, kill = _poly_clip(ax, ay, dm, cand, cand, 0.0, True)
    edge = np.zeros(n, bool)
    edge[cand] = ~kill & (hi - lo > POLY_TOL_EDGE)
    return edge


# ---- SH-MPC scenario sampler: host mirror of tmpc_sample_scenarios_kernel, bit for bit -------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _smp_mix(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _smp_uniform(key, ctr):
    with np.errstate(over="ignore"):
        r = _smp_mix(np.uint64(key) + (np.asarray(ctr, np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    return ((r >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _det_log(x):
    """Natural logarithm from +, x, / only, in the operation order of the device's det_log (no library call: reproducible)."""
    m, e = np.frexp(x)
    small = m < 0.70710678118654752
    m = np.where(small, m * 2.0, m); e = np.where(small, e - 1, e)
    f = (m - 1.0) / (m + 1.0); w = f * f
    p = np.full_like(f, 1.0 / 19.0)
    for c in (17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * w + 1.0 / c
    p = p * w + 1.0
    return e.astype(np.float64) * 0.69314718055994531 + 2.0 * f * p


def _smp_normal(u):
    """Inverse normal CDF (Acklam's rational approximation), same polynomials and operation order as the device."""
    a = (-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00)
    b = (-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01)
    c = (-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00)
    dd = (7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00)
    u = np.asarray(u, np.float64)
    lo = 0.02425

    def tail(p):
        q = np.sqrt(-2.0 * _det_log(p))
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((dd[0] * q + dd[1]) * q + dd[2]) * q + dd[3]) * q + 1.0)

    q = u - 0.5; r = q * q
    mid = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        low = tail(np.where(u < lo, u, 0.5)); high = -tail(np.where(u > 1.0 - lo, 1.0 - u, 0.5))
    return np.where(u < lo, low, np.where(u > 1.0 - lo, high, mid))


def sample_scenarios(pred, prob, n_scenarios, seed):
    """Scenario sampler of SH-MPC (scenario_constraints.cpp:121-131; the scenario_module that implements it is absent -- restated from
    the call: IntegrateAndTranslateToMeanAndVariance): per solver q, obstacle m and scenario s a mode of the obstacle's Gaussian
    mixture is drawn from prob [Q][M][n_modes] and ONE standard-normal pair places the obstacle on every prediction step of that
    mode, o_k = mean_k + R(angle_k) (major_k xi1, minor_k xi2); pred [Q][M][n_modes][N][6] = (x, y, cos angle, sin angle, major, minor).
    Returns samples [Q][N][M * n_scenarios][2] -- equal bit for bit to tmpc_sample_scenarios (counter-based splitmix64 bits, inverse
    normal CDF from +, x, /, sqrt only)."""
    pred = np.asarray(pred, np.float64); prob = np.asarray(prob, np.float64)
    Q, M, n_modes, N, _ = pred.shape
    S = int(n_scenarios)
    out = np.zeros((Q, N, M * S, 2))
    mm, ss = np.meshgrid(np.arange(M), np.arange(S), indexing="ij")
    ctr = ((mm.astype(np.uint64) * np.uint64(S) + ss.astype(np.uint64)) * np.uint64(4))
    for q in range(Q):
        with np.errstate(over="ignore"):
            key = _smp_mix(np.uint64(seed) ^ _smp_mix(np.uint64(q) + np.uint64(0x51ED270B1)))
        um = _smp_uniform(key, ctr)
        mode = np.full((M, S), n_modes - 1)
        cum = np.zeros((M, 1)); done = np.zeros((M, S), bool)
        for j in range(n_modes):
            cum = cum + prob[q, :, j:j + 1]
            hit = (um < cum) & ~done
            mode[hit] = j; done |= hit
        xi1 = _smp_normal(_smp_uniform(key, ctr + np.uint64(1))); xi2 = _smp_normal(_smp_uniform(key, ctr + np.uint64(2)))
        e = pred[q][np.arange(M)[:, None], mode]                       # [M][S][N][6]
        a = e[..., 4] * xi1[..., None]; b = e[..., 5] * xi2[..., None]
        ox = (e[..., 0] + e[..., 2] * a) - e[..., 3] * b
        oy = (e[..., 1] + e[..., 3] * a) + e[..., 2] * b
        out[q, :, :, 0] = ox.transpose(2, 0, 1).reshape(N, M * S)
        out[q, :, :, 1] = oy.transpose(2, 0, 1).reshape(N, M * S)
    return out


def scenario_discard(x0, samples, radius, n_discard):
    """Scenario removal (host mirror of tmpc_scenario_discard_kernel): the n_discard scenarios with the smallest clearance
    min over obstacles m, stages k >= 1 of |o_{m,s,k-1} - p_k| - radius  from the guess (lowest scenario index on ties).
    x0 [N+1][nv]; samples [M][S][N][2].  Returns a bool mask [S] (True = discarded); they count into the bound:
    scenario_risk(S, support, removed=n_discard)."""
    M, S, N, _ = samples.shape
    p = x0[1:N, [IDX["x"], IDX["y"]]]                                  # stages 1..N-1 use prediction steps 0..N-2
    d = samples[:, :, :N - 1, :] - p[None, None]
    clear = (np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) - radius).min(axis=(0, 2))
    out = np.zeros(S, bool)
    out[np.argsort(clear, kind="stable")[:n_discard]] = True
    return out


def scenario_halfspaces(x0, samples, radius, n_rows=24, return_index=False, discard=None):
    """Per-stage polygon construction of SH-MPC (scenario_constraints.cpp:47,76-79 hand this to the external scenario_module, whose
    source is not in the reference tree; restated from the method the reference cites, README.md:22: every sampled obstacle
    position o of stage k gives the halfspace a = (o - p)/|o - p|, b = a.o - radius linearised around the previous plan's position
    p -- the linearisation LinearizedConstraints uses too, linearized_constraints.cpp:84-105 --, the free region of the stage is
    their intersection polygon, and only the halfspaces that form its boundary are constraints of the optimisation).  Exact here:
    `polygon_edges` keeps precisely the non-redundant halfspaces; if the polygon has more than n_rows edges (the solver's capacity)
    the n_rows closest to p are kept (lowest sample index on ties), unused slots stay dummies.  If the halfspaces contradict each
    other (the guess sits in the overlap of inflated discs on opposite sides: an EMPTY polygon) there is no edge; leaving the stage
    unconstrained would certify the most dangerous geometry as safe, so such a stage keeps the n_rows closest halfspaces of all samples
    (contradictory rows: the QP is infeasible or pays slack) and is reported in `empty` (advisor finding, round 2).
    x0 [N+1][nv]; samples [M][S_cen][N][2] (index k-1 for stage k).  Returns a1, a2, b [N][n_rows] with NaN = dummy; rows in order of
    increasing distance.  return_index: also the flat sample index m * S_cen + s behind each row ([N][n_rows], -1 = dummy) and
    empty [N] (bool: the stage's polygon was empty).  discard: bool [S_cen], scenarios left out (scenario_discard)."""
    N = x0.shape[0] - 1
    a1 = np.full((N, n_rows), np.nan); a2 = np.full((N, n_rows), np.nan); b = np.full((N, n_rows), np.nan)
    which = np.full((N, n_rows), -1, np.int32)
    empty = np.zeros(N, bool)
    keep = None
    if discard is not None:                                            # discarded scenarios (scenario_discard) do not exist for this trajectory
        keep = np.flatnonzero(~np.tile(np.asarray(discard, bool), samples.shape[0]))
    for k in range(1, N):
        p = x0[k, [IDX["x"], IDX["y"]]]
        o = samples[:, :, k - 1, :].reshape(-1, 2)
        if keep is not None:
            o = o[keep]
        diff = o - p
        dist = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
        ax = diff[:, 0] / dist; ay = diff[:, 1] / dist
        dm = dist - radius
        idx = np.nonzero(polygon_edges(ax, ay, dm))[0]
        if len(idx) == 0 and len(dm) > 0:                                  # empty polygon: the closest halfspaces of all samples
            idx = np.arange(len(dm)); empty[k] = True
        idx = idx[np.argsort(dm[idx], kind="stable")][:n_rows]             # closest first; stable: lowest sample index on ties
        m = len(idx)
        a1[k, :m] = ax[idx]; a2[k, :m] = ay[idx]
        b[k, :m] = ax[idx] * o[idx, 0] + ay[idx] * o[idx, 1] - radius
        which[k, :m] = idx if keep is None else keep[idx]
    return (a1, a2, b, which, empty) if return_index else (a1, a2, b)


def scenario_support(xtraj, params, pm, row_sample, n_scenarios, tol=1e-6, prefix="disc_0_scenario_constraint"):
    """Support of one trajectory's solution (host mirror of tmpc_scenario_support_kernel; ScenarioSolver::support,
    scenario_constraints.h:38-40, filled by the absent scenario_module -- restated from the definition of the method the reference
    cites, README.md:22): the scenarios with an active constraint  a.p_disc - (b + slack) >= -tol  at the solution, scenario of
    flat sample index i = i % n_scenarios (one scenario = one joint draw of all obstacles over the horizon).
    xtraj [N+1][nx(+1)], params [N][npar], row_sample [N][n_rows].  Returns (support, active_rows)."""
    N, n_rows = row_sample.shape
    off = params[:, pm.index("ego_disc_0_offset")]
    slack = xtraj[:N, 5] if xtraj.shape[1] > 5 else np.zeros(N)
    px = xtraj[:N, 0] + off * np.cos(xtraj[:N, 2]); py = xtraj[:N, 1] + off * np.sin(xtraj[:N, 2])
    active = set(); rows = 0
    for k in range(1, N):
        for r in range(n_rows):
            if row_sample[k, r] < 0:
                continue
            a1, a2, b = (params[k, pm.index(f"{prefix}_{r}_{f}")] for f in ("a1", "a2", "b"))
            if a1 * px[k] + a2 * py[k] - (b + slack[k]) >= -tol:
                active.add(int(row_sample[k, r]) % n_scenarios); rows += 1
    return len(active), rows


def scenario_risk(n_samples, support, confidence=1e-6, removed=0):
    """Bound on the collision probability of a plan certified by a scenario program with `n_samples` scenarios whose solution has
    `support` scenarios of support after `removed` scenarios were discarded (non-convex scenario optimisation, Campi-Garatti-Ramponi
    2018, Theorem 1, the bound SH-MPC builds on -- README.md:22; the discarded scenarios count into the compression set):
        eps(k) = 1 - (beta / (S * C(S, k)))^(1 / (S - k)),  k = support + removed,   eps(S) = 1,
    holds with probability >= 1 - beta (confidence = beta).  The reference's `probabilistic.risk` (settings.yaml) is the value this
    has to stay below."""
    from math import lgamma, log, exp
    S, k = int(n_samples), int(support) + int(removed)
    if k >= S:
        return 1.0
    log_binom = lgamma(S + 1) - lgamma(k + 1) - lgamma(S - k + 1)
    return 1.0 - exp((log(confidence) - log(S) - log_binom) / (S - k))


def scenario_sample_size(risk, confidence=1e-6, max_support=8, removed=0):
    """Smallest number of scenarios S for which a solution with at most `max_support` scenarios of support (after `removed`
    discarded ones) certifies  P(collision) <= risk  with confidence 1 - beta: the smallest S with scenario_risk(S, max_support)
    <= risk (eps decreases in S for fixed k)."""
    lo = max_support + removed + 1
    hi = lo
    while scenario_risk(hi, max_support, confidence, removed) > risk:
        hi *= 2
    while lo < hi:
        mid = (lo + hi) // 2
        if scenario_risk(mid, max_support, confidence, removed) <= risk:
            hi = mid
        else:
            lo = mid + 1
    return lo


def initialize_with_forward_propagation(state, N, dt, nv=NV):
    """Main-solver warm start: constant-velocity forward propagation of the current state (stand-in for
    the previous tick's solution used by initializeWarmstart, acados_solver_interface.cpp:344-376; same
    recursion as initializeWithBraking :303-342 with a = 0)."""
    x0 = np.zeros((N + 1, nv))
    x, y, psi, v, s = state[:5]
    for k in range(N + 1):
        x0[k, :NV] = [0.0, 0.0, x, y, psi, v, s]            # a slack state (column 7) is never initialised: 0
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


def initialize_warmstart(x0, state, xtraj_prev, utraj_prev, shift_previous_solution_forward=True):
    """Solver::initializeWarmstart (acados_solver_interface.cpp:344-376) on one solver's warm start x0 [N+1][nvar],
    in place, from its previous output (xtraj_prev [N+1][nx], utraj_prev [N][nu]).  shift = True:
    [state, out_2, ..., out_{N-1}, out_{N-1}, out_{N-1}]; the inputs of node 0 are written as 0 (the reference reads
    State::get(<input name>) there, an out-of-bounds index: state.cpp:21-24).  shift = False: x0[k] = out_k, k < N."""
    N = x0.shape[0] - 1
    nx = xtraj_prev.shape[1]
    if shift_previous_solution_forward:
        x0[0, :NU] = 0.0
        x0[0, NU:NU + nx] = state[:nx]
        for k in range(1, N + 1):
            ko = N - 1 if k >= N - 1 else k + 1
            x0[k, :NU] = utraj_prev[ko]
            x0[k, NU:NU + nx] = xtraj_prev[ko]
    else:
        for k in range(N):
            x0[k, :NU] = utraj_prev[k]
            x0[k, NU:NU + nx] = xtraj_prev[k]
    return x0


def initialize_with_braking(state, N, dt, deceleration, nv=NV):
    """Solver::initializeWithBraking (acados_solver_interface.cpp:303-342)."""
    x0 = np.zeros((N + 1, nv))
    x, y, psi, v, spline = state[:5]
    a = -abs(deceleration)
    for k in range(N + 1):
        if k >= 1:
            x += v * dt * np.cos(psi); y += v * dt * np.sin(psi); spline += v * dt
            v += a * dt; v = max(v, 0.0)
        x0[k, :NV] = [a, 0.0, x, y, psi, v, spline]
        x0[k, NV:] = state[5:]
    return x0


def map_guidance_trajectories_to_planners(planner_guidance_ids, topology_classes):
    """GuidanceConstraints::mapGuidanceTrajectoriesToPlanners (guidance_constraints.cpp:192-250), integer bookkeeping
    restated literally.  planner_guidance_ids[p] = result.guidance_ID of planner p's last solve; topology_classes[i] =
    class of guidance trajectory i.  Returns (mapping {i: p}, taken [P], existing_guidance [P]).  Note the reference's
    second loop has no `break`: the first unmatched trajectory claims every free planner (its mapping ends at the last
    one) and later unmatched trajectories get none."""
    P = len(planner_guidance_ids)
    taken = [False] * P; existing = [False] * P; mapping = {}; remaining = []
    for i, cls in enumerate(topology_classes):
        found = False
        for p in range(P):
            if planner_guidance_ids[p] == cls and not taken[p]:
                mapping[i] = p; taken[p] = True; existing[p] = True; found = True
                break
        if not found:
            remaining.append(i)
    for i in remaining:
        for p in range(P):
            if not taken[p]:
                mapping[i] = p; taken[p] = True; existing[p] = False
    return mapping, taken, existing
