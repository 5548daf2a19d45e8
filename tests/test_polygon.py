"""the scenario -> polygon construction (mpc_planner_amd/modules.py::polygon_edges / scenario_halfspaces, the host
mirror of tmpc_scenario_halfspaces_kernel).  The scenario_module's source is not in the reference tree (scenario_constraints.cpp:47
only calls it), so the restatement is pinned on the geometry itself: the kept halfspaces must be exactly the non-redundant ones of
the intersection polygon -- checked against scipy's Qhull halfspace intersection and against the unpruned O(n^2) definition."""
import numpy as np
import pytest
from scipy.spatial import HalfspaceIntersection

from mpc_planner_amd import modules as md

P = np.array([1.0, 0.5])
RADIUS = 0.725


def _halfspaces(o, p=P, radius=RADIUS):
    diff = o - p
    dist = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
    return diff[:, 0] / dist, diff[:, 1] / dist, dist - radius, dist


def _brute(ax, ay, dm):
    """The definition without seeds / candidate filter: every halfspace's line clipped by every other halfspace."""
    idx = np.arange(len(dm))
    lo, hi, kill = md._poly_clip(ax, ay, dm, idx, idx, 0.0, True)
    return ~kill & (hi - lo > md.POLY_TOL_EDGE)


def _samples(kind, rng, n=2048):
    if kind == "around":                    # obstacles everywhere: a bounded polygon
        o = P + rng.normal(0, 3.0, (n, 2))
    elif kind == "ahead":                   # four obstacle clouds on one side: an unbounded polygon
        cen = P + np.array([[4, 1], [5, -2], [3, 3], [6, 0.5]], float)[:, None, :]
        o = (cen + rng.normal(0, 0.6, (4, n // 4, 2))).reshape(-1, 2)
    elif kind == "ring":                    # all samples at one distance: every halfspace is an edge
        th = np.sort(rng.uniform(0, 2 * np.pi, 40))
        o = P + 2.0 * np.column_stack([np.cos(th), np.sin(th)])
    elif kind == "few":
        o = P + np.array([[2.0, 0.0], [2.5, 0.0], [-1.5, 1.0]])
    return o[np.linalg.norm(o - P, axis=1) > 0.9]


@pytest.mark.parametrize("seed", range(6))
def test_edges_are_qhulls_nonredundant_halfspaces(seed):
    rng = np.random.default_rng(seed)
    ax, ay, dm, dist = _halfspaces(_samples("around", rng))
    edges = np.nonzero(md.polygon_edges(ax, ay, dm))[0]
    H = HalfspaceIntersection(np.column_stack([ax, ay, -(ax * P[0] + ay * P[1] + dm)]), P)
    assert sorted(set(np.concatenate(H.dual_facets).tolist())) == edges.tolist()
    # and the polygon they bound is the polygon of all halfspaces: its vertices satisfy every halfspace
    v = H.intersections - P
    assert (v @ np.vstack([ax, ay]) <= dm[None, :] + 1e-9).all()


@pytest.mark.parametrize("kind", ["around", "ahead", "ring", "few"])
def test_candidate_filter_changes_nothing(kind):
    for seed in range(4):
        ax, ay, dm, dist = _halfspaces(_samples(kind, np.random.default_rng(100 + seed)))
        assert np.array_equal(md.polygon_edges(ax, ay, dm), _brute(ax, ay, dm)), (kind, seed)


def test_ring_all_edges_and_duplicates_keep_lowest_index():
    rng = np.random.default_rng(5)
    o = _samples("ring", rng)
    ax, ay, dm, dist = _halfspaces(o)
    assert md.polygon_edges(ax, ay, dm).all()
    o2 = np.concatenate([o, o[:5]])                       # exact duplicates of samples 0..4
    ax, ay, dm, dist = _halfspaces(o2)
    e = md.polygon_edges(ax, ay, dm)
    assert e[:len(o)].all() and not e[len(o):].any()


def test_parallel_halfspaces():
    # two samples in the same direction: the farther one is redundant; opposite directions: both are edges of a strip
    o = P + np.array([[2.0, 0.0], [3.0, 0.0], [-2.5, 0.0]])
    ax, ay, dm, dist = _halfspaces(o)
    assert md.polygon_edges(ax, ay, dm).tolist() == [True, False, True]


def test_rows_closest_first_truncated_to_capacity_and_dummies():
    N = 4
    rng = np.random.default_rng(9)
    x0 = np.zeros((N + 1, 7)); x0[:, 2] = P[0]; x0[:, 3] = P[1]
    ring = _samples("ring", rng)                          # 40 edges > 24 rows
    smp = np.zeros((1, len(ring), N, 2)); smp[0, :, :, :] = ring[:, None, :]
    smp[0, :, :, 0] += 1e-3 * np.arange(len(ring))[:, None]          # distinct distances
    a1, a2, b = md.scenario_halfspaces(x0, smp, RADIUS, 24)
    assert np.isnan(a1[0]).all()                          # stage 0: dummies
    for k in range(1, N):
        assert not np.isnan(a1[k]).any()
        margin = b[k] - (a1[k] * P[0] + a2[k] * P[1])
        assert (np.diff(margin) >= 0).all()
        ax, ay, dm, dist = _halfspaces(smp[0, :, k - 1, :])
        assert np.allclose(margin, np.sort(dm)[:24], rtol=0, atol=1e-12)
    few = _samples("few", rng)
    smp = np.zeros((1, len(few), N, 2)); smp[0] = few[:, None, :]
    a1, a2, b = md.scenario_halfspaces(x0, smp, RADIUS, 24)
    assert (~np.isnan(a1[1])).sum() == 2 and np.isnan(a1[1, 2:]).all()          # (2.5, 0) hides behind (2.0, 0)


def test_guess_inside_an_inflated_disc():
    """A warm start that collides with a sample (negative margin): the halfspace excludes the guess itself and is the closest edge."""
    rng = np.random.default_rng(11)
    o = np.concatenate([_samples("around", rng, 256), P[None] + [[0.5, 0.0]]])
    ax, ay, dm, dist = _halfspaces(o)
    e = md.polygon_edges(ax, ay, dm)
    assert dm[-1] < 0 and e[-1] and np.array_equal(e, _brute(ax, ay, dm))


def test_scenario_risk_and_sample_size():
    """The risk bound the support bookkeeping feeds (Campi-Garatti-Ramponi 2018 Thm 1): monotone in the support and in the sample
    size, eps(S) = 1, discarded scenarios count like support, and scenario_sample_size is its inverse."""
    import math
    S, beta = 2048, 1e-6
    eps = [md.scenario_risk(S, k, beta) for k in range(0, 40)]
    assert all(b > a for a, b in zip(eps, eps[1:])) and 0 < eps[0] < eps[-1] < 1
    assert md.scenario_risk(S, S, beta) == 1.0 and md.scenario_risk(S, S + 3, beta) == 1.0
    assert md.scenario_risk(S, 5, beta, removed=3) == md.scenario_risk(S, 8, beta)
    assert md.scenario_risk(4 * S, 8, beta) < md.scenario_risk(S, 8, beta)
    # closed form for k = 0: eps = 1 - (beta / S)^(1/S)
    assert abs(md.scenario_risk(100, 0, 1e-3) - (1 - (1e-3 / 100) ** (1 / 100))) < 1e-14
    # k = 1: C(S, 1) = S
    assert abs(md.scenario_risk(50, 1, 1e-2) - (1 - (1e-2 / (50 * 50)) ** (1 / 49))) < 1e-14
    for risk, n_bar, R in ((0.05, 8, 0), (0.05, 8, 4), (0.01, 5, 0), (0.1, 20, 2)):
        n = md.scenario_sample_size(risk, beta, n_bar, R)
        assert md.scenario_risk(n, n_bar, beta, R) <= risk < md.scenario_risk(n - 1, n_bar, beta, R)


def test_scenario_support_counts_distinct_active_scenarios():
    """Host mirror of tmpc_scenario_support: a plan pressed against two samples of the same scenario and one of another has
    support 2 with 3 active rows; backing off by more than the tolerance empties it; the slack relaxes the rows."""
    from mpc_planner_amd.parameters import define_parameters
    N, S_cen = 4, 8
    pm = define_parameters(5, 0, guidance=False, slack=True, ellipsoids=False, n_scenario=24)
    # samples [M = 2][S_cen][N][2]: obstacle 0 ahead (+x), obstacle 1 to the left (+y); scenario 3 of both is the closest
    smp = np.zeros((2, S_cen, N, 2))
    smp[0, :, :, 0] = 3.0 + 0.1 * np.arange(S_cen)[:, None]; smp[0, 3, :, 0] = 2.0
    smp[1, :, :, 1] = 3.0 + 0.1 * np.arange(S_cen)[:, None]; smp[1, 3, :, 1] = 2.0
    smp[1, 5, :, :] = [-2.5, 0.0]                              # scenario 5: an obstacle 1 sample behind the robot
    x0 = np.zeros((N + 1, 8))
    a1, a2, b, which, empty = md.scenario_halfspaces(x0, smp, RADIUS, 24, return_index=True)
    assert not empty.any()
    assert sorted(which[1][which[1] >= 0].tolist()) == [3, S_cen + 3, S_cen + 5]
    params = np.zeros((N, pm.length()))
    md.halfspace_rows_set_parameters(pm, params, 0.0, (a1, a2, b), "disc_0_scenario_constraint", 24)
    xtraj = np.zeros((N + 1, 6))
    assert md.scenario_support(xtraj, params, pm, which, S_cen) == (0, 0)
    xtraj[1:, 0] = 2.0 - RADIUS; xtraj[1:, 1] = 2.0 - RADIUS        # in the corner of the polygon: both scenario-3 rows active
    assert md.scenario_support(xtraj, params, pm, which, S_cen) == (1, 2 * (N - 1))
    xtraj[2, 0] = -2.5 + RADIUS; xtraj[2, 1] = 0.0                 # stage 2 against the sample behind instead
    assert md.scenario_support(xtraj, params, pm, which, S_cen) == (2, 2 * (N - 2) + 1)
    xtraj[:, 5] = 0.05                                            # a positive slack relaxes every row by 5 cm
    assert md.scenario_support(xtraj, params, pm, which, S_cen) == (0, 0)
    assert md.scenario_support(xtraj, params, pm, which, S_cen, tol=0.06) == (2, 2 * (N - 2) + 1)


def test_random_geometries_against_the_definition():
    """200 seeded sample sets of mixed shape (3 ... 400 samples; clouds, lines of samples, samples on both sides, guesses close to a
    cloud): the filtered edge set equals the unpruned definition, and the kept rows are sorted by margin."""
    rng = np.random.default_rng(2024)
    n_nonempty = 0
    for case in range(200):
        n = int(rng.integers(3, 400))
        kind = case % 4
        if kind == 0:
            o = P + rng.normal(0, rng.uniform(1.0, 6.0), (n, 2))
        elif kind == 1:                                             # samples along a line (a wall of predictions)
            t = rng.uniform(-6, 6, n)
            th = rng.uniform(0, np.pi)
            o = P + np.array([3.0 * np.cos(th + 1.3), 3.0 * np.sin(th + 1.3)]) + np.outer(t, [np.cos(th), np.sin(th)]) + rng.normal(0, 0.02, (n, 2))
        elif kind == 2:                                             # two clouds on opposite sides
            o = P + np.where(rng.random((n, 1)) < 0.5, 1.0, -1.0) * np.array([4.0, 0.5]) + rng.normal(0, 0.8, (n, 2))
        else:                                                       # a cloud the guess almost touches
            o = P + np.array([1.2, 0.0]) + rng.normal(0, 0.3, (n, 2))
        o = o[np.linalg.norm(o - P, axis=1) > 0.05]
        if len(o) < 2:
            continue
        ax, ay, dm, _ = _halfspaces(o)
        e = md.polygon_edges(ax, ay, dm)
        assert np.array_equal(e, _brute(ax, ay, dm)), case
        n_nonempty += bool(e.any())
    assert n_nonempty >= 150          # (a guess inside overlapping discs on opposite sides has an EMPTY polygon: no edge, all rows dummies)


def test_empty_polygon_keeps_the_closest_halfspaces_and_is_reported():
    """Advisor (round 2): a guess inside the overlap of inflated discs on opposite sides has contradictory halfspaces -- an empty
    polygon.  The stage must not be left unconstrained (that would certify the most dangerous geometry as safe): it keeps the n_rows
    closest halfspaces (infeasible rows: the QP fails or pays slack) and the stage is flagged."""
    N, S_cen = 3, 6
    smp = np.zeros((2, S_cen, N, 2))
    smp[0, :, :, 0] = 0.3 + 0.01 * np.arange(S_cen)[:, None]          # obstacle 0: 0.3 m ahead -> margins -0.4 (inside the inflated disc)
    smp[1, :, :, 0] = -0.3 - 0.01 * np.arange(S_cen)[:, None]         # obstacle 1: 0.3 m behind: contradicts obstacle 0
    x0 = np.zeros((N + 1, 8))
    a1, a2, b, which, empty = md.scenario_halfspaces(x0, smp, RADIUS, 4, return_index=True)
    assert empty[1:].all() and not empty[0]
    assert which[1].tolist() == [0, S_cen, 1, S_cen + 1]                 # closest first, lowest sample index on ties
    assert np.isfinite(a1[1]).all() and (b[1] < 0).all()                 # rows that exclude the guess itself
    # a feasible neighbour geometry is unaffected
    smp[1, :, :, 0] = -3.0
    smp[0, :, :, 0] = 3.0
    _, _, _, which2, empty2 = md.scenario_halfspaces(x0, smp, RADIUS, 4, return_index=True)
    assert not empty2.any() and (which2[1] >= 0).sum() == 2
