"""Deterministic synthetic Jackal T-MPC scenes.

seed = 1000 + scene_idx; weights/geometry from mpc_planner_jackalsimulator/config/settings.yaml:33-40,75-89.
A scene = one control tick: one initial state, one reference path, M obstacles with constant-velocity
predictions, and a batch of B guidance trajectories (+1 non-guided T-MPC++ planner if requested), each of
which becomes one independent NLP (guidance_constraints.cpp:279-361).
"""
import numpy as np

from . import modules as md
from .parameters import define_parameters

WEIGHTS = dict(acceleration=0.34, angular_velocity=0.85, velocity=0.55, reference_velocity=2.0,
               contour=0.05, lag=0.75, terminal_angle=100.0, terminal_contouring=10.0,
               slack=10000.0)        # slack-model configurations only (mpc_planner_rosnavigation/config/settings.yaml:86)
ROBOT_RADIUS = 0.325
OBSTACLE_RADIUS = 0.4
DT = 0.2
A_MAX = 4.0   # lateral amplitudes of the guidance set [m] (linspace(-4, 4, B)); candidates that the model cannot
              # follow (|w| <= 0.8, |a| <= 2, v <= 3: solver_model.py:204-205) or that hit a prediction are re-drawn
W_LIMIT, A_LIMIT, V_LIMIT = 0.7, 1.8, 2.8     # reachability margins inside the model's input / speed bounds


def _smoothstep(tau):
    """Quintic step 0 -> 1 with zero velocity and acceleration at both ends."""
    tau = np.clip(tau, 0.0, 1.0)
    return tau ** 3 * (10.0 - 15.0 * tau + 6.0 * tau * tau)


def _smoothstep_d(tau):
    tau = np.clip(tau, 0.0, 1.0)
    return 30.0 * tau * tau * (1.0 - tau) ** 2


def guidance_candidates(state, t, A, v_cruise, t_acc, shape):
    """Guidance trajectories (position / velocity at t = k dt; arrays over candidates: A, v_cruise, t_acc, shape [K] ->
    gpos, gvel [K][N+1][2]) the way a guidance planner's goal grid produces them (guidance_planner is external; SURVEY 8d
    prescribes lateral profiles of amplitude A): the robot accelerates from its current speed to `v_cruise` within `t_acc` and
    moves laterally by A with a quintic step (shape 0: ends on a goal offset by A from the path, like the lateral goal grid of
    guidance_planner.yaml) or leaves and rejoins the path (shape 1).  Tangential to the current heading at t = 0, so it is
    consistent with xinit."""
    A = np.atleast_1d(np.asarray(A, float))[:, None]; vc = np.atleast_1d(np.asarray(v_cruise, float))[:, None]
    shape = np.atleast_1d(np.asarray(shape))[:, None]
    T = t[-1]; tt = t[None, :]
    v0 = state[3]
    ta = np.minimum(np.atleast_1d(np.asarray(t_acc, float))[:, None], T)
    vx = v0 + (vc - v0) * np.minimum(tt / ta, 1.0)
    xs = np.where(tt < ta, v0 * tt + 0.5 * (vc - v0) * tt * tt / ta, v0 * ta + 0.5 * (vc - v0) * ta + vc * (tt - ta))
    tau = tt / T
    up = tau <= 0.5
    y = A * np.where(shape == 0, _smoothstep(tau), np.where(up, _smoothstep(2.0 * tau), _smoothstep(2.0 - 2.0 * tau)))
    vy = A * np.where(shape == 0, _smoothstep_d(tau) / T,
                      np.where(up, _smoothstep_d(2.0 * tau), -_smoothstep_d(2.0 - 2.0 * tau)) * 2.0 / T)
    return np.stack([xs, y], 2), np.stack([vx, vy], 2)


def guidance_reachable(gpos, gvel, state):
    """Can the unicycle follow the guidance from `state` within its bounds (solver_model.py:204-205)?  Heading rate, speed and
    acceleration of the guidance at the nodes, with margins.  gvel [K][N+1][2] -> bool [K]."""
    psi = np.unwrap(np.arctan2(gvel[..., 1], gvel[..., 0]), axis=-1)
    v = np.hypot(gvel[..., 0], gvel[..., 1])
    w = np.diff(psi, axis=-1) / DT; a = np.diff(v, axis=-1) / DT
    return ((np.abs(w).max(-1) <= W_LIMIT) & (np.abs(a).max(-1) <= A_LIMIT) & (v.max(-1) <= V_LIMIT) & (v.min(-1) >= 0.05)
            & (np.abs(psi[..., 0] - state[2]) < 0.1))


def reference_path_segments(rng, S=5, seg_len=6.0):
    """S cubic segments of 6 m along +x with a lateral sine, (a,b,c,d,start) per contouring.cpp:94-124."""
    amp = rng.uniform(0.0, 1.0)
    wavelength = rng.uniform(18.0, 36.0)
    phase = rng.uniform(0.0, 2 * np.pi)
    f = lambda x: amp * (np.sin(2 * np.pi * x / wavelength + phase) - np.sin(phase))
    df = lambda x: amp * 2 * np.pi / wavelength * np.cos(2 * np.pi * x / wavelength + phase)
    segs = np.zeros((S, 9))
    for i in range(S):
        x0, x1, L = seg_len * i, seg_len * (i + 1), seg_len
        y0, y1, m0, m1 = f(x0), f(x1), df(x0), df(x1)
        # cubic Hermite in t = s - start
        d = y0; c = m0
        b = (3 * (y1 - y0) / L - 2 * m0 - m1) / L
        a = (m0 + m1 - 2 * (y1 - y0) / L) / (L * L)
        segs[i] = [0.0, 0.0, 1.0, x0, a, b, c, d, x0]
    return segs


def decomp_corridor(rng, segs, state, N, n_rows=12):
    """Stand-in for DecompConstraints::update (decomp_constraints.cpp:53-118; DecompUtil + costmap absent): per stage
    k >= 1 a convex polytope around the reference-path point at s_k = s + k v dt: two corridor walls parallel to the
    path, a front and a back cap and four diagonal cuts (8 rows); the remaining slots stay dummies like the
    reference's padding (:106-111).  Returns a1, a2, b [N][n_rows] with NaN = dummy."""
    a1 = np.full((N, n_rows), np.nan); a2 = np.full((N, n_rows), np.nan); b = np.full((N, n_rows), np.nan)
    half_w = rng.uniform(2.2, 3.5, 2)                         # left / right wall distance [m]
    reach = rng.uniform(5.0, 7.0)
    s = state[4]
    for k in range(N - 1):
        i = min(int(s // 6.0), segs.shape[0] - 1); t = s - segs[i, 8]
        px = ((segs[i, 0] * t + segs[i, 1]) * t + segs[i, 2]) * t + segs[i, 3]
        py = ((segs[i, 4] * t + segs[i, 5]) * t + segs[i, 6]) * t + segs[i, 7]
        dx = (3 * segs[i, 0] * t + 2 * segs[i, 1]) * t + segs[i, 2]
        dy = (3 * segs[i, 4] * t + 2 * segs[i, 5]) * t + segs[i, 6]
        nrm = np.hypot(dx, dy); tx, ty = dx / nrm, dy / nrm
        normals = [(-ty, tx, half_w[0]), (ty, -tx, half_w[1]), (tx, ty, reach), (-tx, -ty, reach)]
        for sx, sy in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
            nx_, ny_ = (sx * tx - sy * ty) / np.sqrt(2.0), (sx * ty + sy * tx) / np.sqrt(2.0)
            normals.append((nx_, ny_, 0.8 * (reach + half_w.mean())))
        for r, (ax, ay, dist) in enumerate(normals):
            a1[k + 1, r] = ax; a2[k + 1, r] = ay; b[k + 1, r] = ax * px + ay * py + dist
        s += state[3] * DT
    return a1, a2, b


def scenario_samples(rng, pos0, vel, N, n_samples):
    """Gaussian-mixture obstacle scenarios (3 modes: straight / veer left / veer right, weights .5/.25/.25) with
    process noise integrated along the horizon -- stand-in for the scenario_module's sampler
    (scenario_constraints.cpp:125-131 IntegrateAndTranslateToMeanAndVariance; source absent).
    Returns samples [M][n_samples][N][2]."""
    M = pos0.shape[0]
    mode = rng.choice(3, size=(M, n_samples), p=[0.5, 0.25, 0.25])
    turn = np.array([0.0, 0.06, -0.06])[mode]
    c, s = np.cos(turn), np.sin(turn)
    v = np.stack([c * vel[:, None, 0] - s * vel[:, None, 1], s * vel[:, None, 0] + c * vel[:, None, 1]], -1)   # [M][S][2]
    steps = np.arange(N)[None, None, :, None]
    noise = np.cumsum(rng.normal(0.0, 0.05 * DT, (M, n_samples, N, 2)), axis=2)
    return pos0[:, None, None, :] + v[:, :, None, :] * DT * steps + noise


def mixture_prediction(obs_pos, n_extra=0, dt=DT):
    """The three-mode Gaussian-mixture prediction of scenario_samples (straight / veer left / veer right) as the tensor the device
    sampler reads (tmpc_sample_scenarios): [M][3][N + n_extra][6] = (x, y, cos angle, sin angle, major, minor) per obstacle, mode and
    prediction step; radii = integrated standard deviations growing along the horizon (data_preparation.cpp:170-186).
    obs_pos [M][N][2]: the obstacles' mean predictions (make_scene()["obstacles"]["pos"])."""
    M, N, _ = obs_pos.shape
    T = N + n_extra
    vel = (obs_pos[:, 1] - obs_pos[:, 0]) / dt
    steps = np.arange(T)
    out = np.zeros((M, 3, T, 6))
    for j, turn in enumerate((0.0, 0.06, -0.06)):
        c, s_ = np.cos(turn), np.sin(turn)
        v = np.stack([c * vel[:, 0] - s_ * vel[:, 1], s_ * vel[:, 0] + c * vel[:, 1]], 1)
        out[:, j, :, 0:2] = obs_pos[:, :1, :] + v[:, None, :] * dt * steps[None, :, None]
        ang = np.arctan2(v[:, 1], v[:, 0])
        out[:, j, :, 2] = np.cos(ang)[:, None]; out[:, j, :, 3] = np.sin(ang)[:, None]
        out[:, j, :, 4] = 0.05 * dt * np.sqrt(steps + 1.0)[None]        # along-track
        out[:, j, :, 5] = 0.03 * dt * np.sqrt(steps + 1.0)[None]        # cross-track
    return out


MIXTURE_WEIGHTS = (0.5, 0.25, 0.25)


def make_scene(scene_idx, N=20, M=8, B=64, S=5, tmpc_pp=False, gaussian=False, guidance=True, slack=False,
               n_decomp=0, n_scenario=0, n_samples=256, chance=False, inside_share=0.0):
    """Returns dict(xinit [Bt][nx], x0 [Bt][N+1][nv], params [Bt][N][npar], pm, guidance_id [Bt]); nx = 5, nv = 7, or
    6 / 8 with the slack model.  n_scenario > 0 builds the SH-MPC problem (cfg 5): no ellipsoid / topology rows, 24
    scenario halfspaces per stage from M obstacles x n_samples scenarios, one set per guidance trajectory.
    inside_share > 0 (round-4 verdict, next-8): that share of the guidance trajectories (own seeded stream: the rest of the scene is bitwise
    what inside_share = 0 gives) gets ONE guidance point moved INSIDE the disc of radius 1e-3 + robot_radius around the nearest obstacle
    prediction -- what a guidance path computed on the previous tick's predictions looks like -- so that LinearizedConstraints::projectToSafety
    (linearized_constraints.cpp:130-148) is NOT the identity there; `inside` [Bt] marks them, `inside_at` [Bt][2] = (stage, obstacle) or (-1, -1)."""
    rng = np.random.Generator(np.random.PCG64(1000 + scene_idx))
    rng_in = np.random.Generator(np.random.PCG64(770000 + scene_idx))
    ellipsoids = n_scenario == 0 and not chance         # chance: GaussianConstraintModule instead of the ellipsoids (jackal default)
    if chance:
        gaussian = True
    if n_scenario:
        guidance = False
    pm = define_parameters(S, M, guidance=guidance, slack=slack, ellipsoids=ellipsoids, n_scenario=n_scenario,
                           n_decomp=n_decomp, gaussian=chance)
    npar = pm.length()
    nx, nv = 5 + int(slack), 7 + int(slack)
    state = np.array([0.0, 0.0, 0.0, rng.uniform(0.5, 2.0), 0.0] + [0.0] * int(slack))   # x,y,psi,v,spline(,slack)
    segs = reference_path_segments(rng, S)
    # obstacles: constant velocity (data_preparation.cpp:58-79): mode[i] = pos0 + vel*dt*i.
    # Half of them are "crossing pedestrians" timed to meet the robot's nominal progress along the path
    # (so that topology / collision rows are active at the optimum, like the reference's
    # pedestrian_simulator scenarios); the rest are background obstacles from the SURVEY 8d box.
    speed = rng.uniform(0.6, 1.6, M)
    heading = np.where(rng.uniform(size=M) < 0.5, 1.0, -1.0) * np.pi / 2 + rng.uniform(-0.5, 0.5, M)
    vel = np.stack([speed * np.cos(heading), speed * np.sin(heading)], 1)
    pos0 = np.stack([rng.uniform(3.0, 18.0, M), rng.uniform(-4.0, 4.0, M)], 1)
    n_cross = (M + 1) // 2
    v_nom = 0.5 * (state[3] + WEIGHTS["reference_velocity"])

    def leaves_room(p0, vl):
        """A tick that starts (almost) in collision has no feasible plan at all: the robot, coasting straight on, must keep
        r_obstacle + r_robot + 0.3 m from the prediction during the first 1.6 s -- time enough for a guidance trajectory to
        pass on either side."""
        i = np.arange(8)
        robot = np.stack([state[3] * DT * (i + 1), np.zeros(8)], 1)
        return np.linalg.norm(p0[None, :] + vl[None, :] * DT * i[:, None] - robot, axis=1).min() >= OBSTACLE_RADIUS + ROBOT_RADIUS + 0.3

    for j in range(M):
        for _try in range(100):
            if j < n_cross:
                x_c = rng.uniform(1.5, 7.0)
                t_c = x_c / v_nom + rng.uniform(-0.5, 0.5)
                y_c = rng.uniform(-0.3, 0.3)
                pos0[j] = np.array([x_c, y_c]) - vel[j] * t_c
            elif _try > 0:
                pos0[j] = [rng.uniform(3.0, 18.0), rng.uniform(-4.0, 4.0)]
            if leaves_room(pos0[j], vel[j]):
                break
    steps = np.arange(N)[None, :, None]
    obs = dict(pos=pos0[:, None, :] + vel[:, None, :] * DT * steps, angle=np.zeros((M, N)),
               radius=np.full(M, OBSTACLE_RADIUS), major=np.zeros((M, N)), minor=np.zeros((M, N)),
               chi=np.ones(M))
    if gaussian:                                                            # data_preparation.cpp:170-186
        acc = np.sqrt(np.cumsum(np.full(N, (0.3 * DT) ** 2)))
        obs["major"][:] = acc; obs["minor"][:] = acc
        obs["chi"][:] = -np.log(0.05) / 0.5                                 # ExponentialQuantile(0.5, 0.95)

    base = np.zeros((N, npar))
    md.mpc_base_set_parameters(pm, base, WEIGHTS)
    md.contouring_set_parameters(pm, base, WEIGHTS, segs)
    if ellipsoids:
        md.ellipsoid_set_parameters(pm, base, state[:2], obs, ROBOT_RADIUS)
    if chance:
        md.gaussian_set_parameters(pm, base, state[:2], obs, ROBOT_RADIUS, risk=0.05)    # settings.yaml probabilistic/risk
    if n_decomp:
        corridor = decomp_corridor(rng, segs, state, N, n_decomp)
        md.halfspace_rows_set_parameters(pm, base, state[0], corridor, "disc_0_decomp", n_decomp)
    samples = scenario_samples(rng, pos0, vel, N, n_samples) if n_scenario else None
    main_x0 = md.initialize_with_forward_propagation(state, N, DT, nv)

    Bt = B + (1 if tmpc_pp else 0)
    xinit = np.tile(state, (Bt, 1))
    x0 = np.zeros((Bt, N + 1, nv)); params = np.zeros((Bt, N, npar))
    guidance_id = np.zeros(Bt, np.int32)
    guidance_pos = np.zeros((B, N + 1, 2)); guidance_vel = np.zeros((B, N + 1, 2))        # what the guidance planner hands over
    inside = np.zeros(Bt, bool); inside_at = np.full((Bt, 2), -1, np.int32)
    T = N * DT; t = np.arange(N + 1) * DT
    v_ref = WEIGHTS["reference_velocity"]
    amps = np.linspace(-A_MAX, A_MAX, B) + rng.normal(0.0, 0.05, B) if B > 1 else np.array([rng.normal(0.0, 0.5)])
    # inflated obstacle radius per prediction step (Gaussian predictions: major sqrt(chi), ellipsoid_constraints.py:94-100)
    rad = obs["major"] * np.sqrt(obs["chi"])[:, None] + OBSTACLE_RADIUS + ROBOT_RADIUS + 0.1          # [M][N]
    r_samples = OBSTACLE_RADIUS + ROBOT_RADIUS + 0.05
    # candidate pool of the scene: the B nominal profiles (amplitudes above, cruise at v_ref) followed by K random ones; a
    # trajectory keeps its nominal profile if the robot can follow it and it clears every prediction, otherwise it takes the
    # next unused pool member that does (what a guidance planner returns are collision-free, dynamically feasible paths)
    K = 1024 if samples is None else 384
    cA = np.concatenate([amps, rng.uniform(-A_MAX, A_MAX, K)])
    cv = np.concatenate([np.full(B, v_ref), rng.uniform(0.6, 2.6, K)])
    cta = np.concatenate([np.full(B, 1.5), rng.uniform(1.0, 3.0, K)])
    csh = np.concatenate([np.zeros(B, int), rng.integers(0, 2, K)])
    cpos, cvel = guidance_candidates(state, t, cA, cv, cta, csh)
    reach = guidance_reachable(cpos, cvel, state)
    if samples is not None:      # SH-MPC: clearance from every sampled scenario, not only from the mean prediction
        cloud = samples[:, :, :N - 1, :].reshape(-1, N - 1, 2)
        clear = np.full(B + K, np.inf)
        for c0 in range(0, B + K, 32):                                   # chunked: [32][points][N-1]
            dd = cpos[c0:c0 + 32, None, 1:N, :] - cloud[None]
            clear[c0:c0 + 32] = np.sqrt((dd * dd).sum(-1)).min(axis=(1, 2)) - r_samples
    else:
        dd = cpos[:, None, 1:N, :] - obs["pos"][None, :, :N - 1, :]
        clear = (np.sqrt((dd * dd).sum(-1)) - rad[None, :, :N - 1]).min(axis=(1, 2))
    if n_decomp:                 # static obstacles: stay inside the free-space polytope of every stage (decomp rows, slack = 0)
        da1, da2, db = corridor
        inside = np.ones(B + K, bool)
        for k in range(1, N):
            live = ~np.isnan(da1[k])
            val = cpos[:, k, 0:1] * da1[k][None, live] + cpos[:, k, 1:2] * da2[k][None, live] - db[k][None, live]
            inside &= (val <= -0.05).all(axis=1)
        clear = np.where(inside, clear, np.minimum(clear, -1e-3))
    good = reach & (clear >= 0.0)
    spare = [int(i) for i in np.nonzero(good[B:])[0] + B]
    fallback = int(np.argmax(np.where(reach, clear, -np.inf))) if reach.any() else 0
    for b in range(B):
        if good[b]:
            pick = b
        elif spare:
            pick = spare.pop(0)
        else:
            pick = fallback
        best = (clear[pick], cpos[pick], cvel[pick])
        _, gpos, gvel = best
        gpos = gpos.copy()
        # stand-in for LinearizedConstraints::projectToSafety (linearized_constraints.cpp:130-148; the
        # Douglas-Rachford projection lives in ros_tools, source absent): if no sampled guidance keeps
        # r + robot_radius from every obstacle prediction, push the offending points radially out so the
        # reference's projection would be the identity on what we hand to the solver.
        r_min = 1e-3 + ROBOT_RADIUS + 1e-6
        for _sweep in range(50 if best[0] < 0.0 else 0):           # (a candidate that clears every prediction needs no push-out)
            moved = False
            for k in range(1, N):
                for j in range(M):
                    o = obs["pos"][j, k - 1]
                    dvec = gpos[k] - o
                    dist = np.sqrt(dvec[0] * dvec[0] + dvec[1] * dvec[1])
                    if dist < r_min:
                        gpos[k] = o + (dvec / dist if dist > 1e-12 else np.array([0.0, 1.0])) * (r_min * 1.001)
                        moved = True
            if not moved:
                break
        if samples is not None:            # same stand-in against every sampled scenario (radius of the scenario rows)
            r_s = OBSTACLE_RADIUS + ROBOT_RADIUS + 1e-3
            for k in range(1, N if best[0] < 0.0 else 1):
                cloud = samples[:, :, k - 1, :].reshape(-1, 2)
                for _sweep in range(60):
                    dvec = gpos[k] - cloud
                    dist = np.sqrt(dvec[:, 0] * dvec[:, 0] + dvec[:, 1] * dvec[:, 1])
                    j = int(np.argmin(dist))
                    if dist[j] >= r_s:
                        break
                    gpos[k] = cloud[j] + (dvec[j] / dist[j] if dist[j] > 1e-12 else np.array([0.0, 1.0])) * (r_s * 1.001)
        if inside_share > 0.0 and guidance and samples is None and rng_in.uniform() < inside_share:
            k_in = int(rng_in.integers(2, N - 1))                        # a stage whose row is built from prediction step k_in - 1
            dist_k = np.hypot(*(gpos[k_in][None, :] - obs["pos"][:, k_in - 1]).T)
            j_in = int(np.argmin(dist_k))
            o = obs["pos"][j_in, k_in - 1]
            ang = np.arctan2(gpos[k_in][1] - o[1], gpos[k_in][0] - o[0]) + rng_in.uniform(-0.6, 0.6)
            gpos[k_in] = o + rng_in.uniform(0.2, 0.9) * (1e-3 + ROBOT_RADIUS) * np.array([np.cos(ang), np.sin(ang)])
            inside[b] = True; inside_at[b] = (k_in, j_in)
        guidance_pos[b] = gpos; guidance_vel[b] = gvel
        x0[b] = md.initialize_solver_with_guidance(main_x0.copy(), gpos, gvel)
        params[b] = base
        if guidance:
            lin = md.linearized_update(x0[b], obs["pos"], ROBOT_RADIUS)
            md.linearized_set_parameters(pm, params[b], state[0], lin, n_rows=M)
        if n_scenario:
            rows = md.scenario_halfspaces(x0[b], samples, OBSTACLE_RADIUS + ROBOT_RADIUS, n_scenario)
            md.halfspace_rows_set_parameters(pm, params[b], state[0], rows, "disc_0_scenario_constraint", n_scenario)
        guidance_id[b] = b
    if tmpc_pp:                                                             # non-guided planner
        x0[B] = main_x0
        params[B] = base
        md.linearized_set_parameters(pm, params[B], state[0], None, n_rows=M)
        guidance_id[B] = 2 * B                                              # guidance_constraints.cpp:349
    return dict(xinit=xinit, x0=x0, params=params, pm=pm, guidance_id=guidance_id, obstacles=obs,
                segments=segs, N=N, M=(M if ellipsoids else 0), S=S, n_lin=(M if guidance else 0), n_gauss=(M if chance else 0),
                n_slk=n_scenario + n_decomp, slack=int(slack), samples=samples, guidance_pos=guidance_pos, guidance_vel=guidance_vel,
                inside=inside, inside_at=inside_at)


def _make_scene_kw(args):
    idx, kw = args
    return make_scene(idx, **kw)


def make_batch(scene_indices, workers=1, **kw):
    """Concatenate scenes into one launch batch (throughput mode: S scenes x B trajectories).  workers > 1 generates the
    scenes in that many forked processes (call it before anything initialises the GPU runtime in this process)."""
    scene_indices = list(scene_indices)
    if workers > 1 and len(scene_indices) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(scene_indices))) as pool:
            scenes = pool.map(_make_scene_kw, [(i, kw) for i in scene_indices], chunksize=max(1, len(scene_indices) // (4 * workers)))
    else:
        scenes = [make_scene(i, **kw) for i in scene_indices]
    out = dict(scenes[0])
    for key in ("xinit", "x0", "params", "guidance_id", "inside", "inside_at"):
        out[key] = np.concatenate([s[key] for s in scenes], 0)
    out["scene_of"] = np.concatenate([np.full(len(s["xinit"]), i, np.int32) for i, s in enumerate(scenes)])
    # what a control tick hands over per scene / per trajectory (the end-to-end step of bench.py uploads exactly these):
    # obstacle predictions [n_scenes][M][N][2], guidance position / velocity at t = k dt [B][N+1][2]
    if all(s.get("guidance_pos") is not None for s in scenes):
        out["guidance_pos"] = np.concatenate([s["guidance_pos"] for s in scenes], 0)
        out["guidance_vel"] = np.concatenate([s["guidance_vel"] for s in scenes], 0)
    out["obstacle_pos"] = np.stack([s["obstacles"]["pos"] for s in scenes])
    return out
