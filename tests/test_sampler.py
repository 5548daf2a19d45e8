"""CPU: the host mirror of the SH-MPC scenario sampler and of the scenario-removal step (f-3) -- the reproducible building blocks
(splitmix64 bits, inverse normal CDF and logarithm from basic operations only) and the statistics of the draws."""
import numpy as np

from mpc_planner_amd import modules as md


def test_reproducible_normal_and_log():
    from scipy.stats import norm
    u = np.concatenate([np.linspace(1e-12, 1 - 1e-12, 100001), 10.0 ** -np.arange(3, 15)])
    assert np.abs(md._smp_normal(u) - norm.ppf(u)).max() < 1e-8           # Acklam: 1.15e-9 relative
    x = np.exp(np.linspace(-60, 20, 20001))
    assert (np.abs(md._det_log(x) - np.log(x)) <= 2e-14 * np.maximum(1.0, np.abs(np.log(x)))).all()
    # the uniform bits: in (0, 1), 53 bits, well spread
    r = md._smp_uniform(np.uint64(42), np.arange(200000, dtype=np.uint64))
    assert r.min() > 0 and r.max() < 1 and abs(r.mean() - 0.5) < 3e-3 and abs(r.var() - 1 / 12) < 1e-3
    assert abs(np.corrcoef(r[:-1], r[1:])[0, 1]) < 0.01


def test_sampler_statistics_and_determinism():
    Q, M, n_modes, N, S = 3, 4, 3, 12, 4096
    rng = np.random.default_rng(1)
    pred = np.zeros((Q, M, n_modes, N, 6))
    pred[..., 0] = rng.normal(0, 3, (Q, M, n_modes, 1)) + 0.2 * np.arange(N)
    pred[..., 1] = rng.normal(0, 3, (Q, M, n_modes, 1))
    ang = rng.uniform(-1, 1, (Q, M, n_modes, 1))
    pred[..., 2] = np.cos(ang); pred[..., 3] = np.sin(ang)
    pred[..., 4] = 0.05 * np.sqrt(np.arange(1, N + 1)); pred[..., 5] = 0.02 * np.sqrt(np.arange(1, N + 1))
    prob = np.tile([0.6, 0.3, 0.1], (Q, M, 1))
    a = md.sample_scenarios(pred, prob, S, 99)
    assert np.array_equal(a, md.sample_scenarios(pred, prob, S, 99))        # a pure function of its arguments
    assert not np.array_equal(a, md.sample_scenarios(pred, prob, S, 100))   # the seed matters
    assert not np.array_equal(a[0], a[1])                                   # and so does the solver index (per-solver draws)
    smp = a.reshape(Q, N, M, S, 2)
    # mode frequencies and per-mode moments: identify the mode of a scenario from its step-0 sample (modes are metres apart)
    for q in range(Q):
        for m in range(M):
            d = np.linalg.norm(smp[q, 0, m][:, None, :] - pred[q, m, :, 0, 0:2][None], axis=2)
            mode = d.argmin(1)
            frac = np.bincount(mode, minlength=3) / S
            assert np.abs(frac - [0.6, 0.3, 0.1]).max() < 0.03
            sel = mode == 0
            k = N - 1
            dev = smp[q, k, m][sel] - pred[q, m, 0, k, 0:2]
            c, s_ = pred[q, m, 0, k, 2], pred[q, m, 0, k, 3]
            along = dev[:, 0] * c + dev[:, 1] * s_; cross = -dev[:, 0] * s_ + dev[:, 1] * c
            assert abs(along.std() / pred[q, m, 0, k, 4] - 1) < 0.08 and abs(cross.std() / pred[q, m, 0, k, 5] - 1) < 0.08
            assert abs(along.mean()) < 0.2 * pred[q, m, 0, k, 4] and abs(cross.mean()) < 0.2 * pred[q, m, 0, k, 5]


def test_scenario_removal_takes_the_most_constraining_scenarios():
    M, S, N = 2, 16, 6
    smp = np.zeros((M, S, N, 2))
    smp[0, :, :, 0] = 5.0 + np.arange(S)[:, None]                 # obstacle 0: scenario s is 5 + s metres ahead
    smp[1, :, :, 1] = 9.0                                          # obstacle 1: far to the side for every scenario ...
    smp[1, 11, 3, :] = [0.4, 0.9]                                  # ... except scenario 11 at one step, close to the plan
    x0 = np.zeros((N + 1, 8))
    mk = md.scenario_discard(x0, smp, 0.725, 3)
    assert np.flatnonzero(mk).tolist() == [0, 1, 11]               # the clearance is the minimum over obstacles AND stages
    a1, a2, b, which, empty = md.scenario_halfspaces(x0, smp, 0.725, 6, return_index=True, discard=mk)
    assert not np.isin(which[which >= 0] % S, [0, 1, 11]).any()
    # ties: lowest scenario index first
    smp[0, :, :, 0] = 5.0
    assert np.flatnonzero(md.scenario_discard(x0, smp, 0.725, 2)).tolist() == [0, 11]     # scenario 11 (closest), then the lowest index of the tie
