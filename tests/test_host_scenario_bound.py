"""Scenario-risk bookkeeping with removed scenarios (modules.scenario_risk / scenario_sample_size; solver.optimize_scenarios' `max_support`):
`max_support` bounds the support AFTER the removal and the removed scenarios enter through `removed` -- a caller that ignores them
(sample size chosen with removed = 0) would certify less than it claims."""
from mpc_planner_amd import modules as md


def test_removed_scenarios_count_into_the_bound():
    risk, n_bar, removed = 0.05, 6, 7
    S0 = md.scenario_sample_size(risk, max_support=n_bar, removed=0)
    S1 = md.scenario_sample_size(risk, max_support=n_bar, removed=removed)
    assert S1 > S0                                                         # removal costs samples
    assert md.scenario_risk(S1, n_bar, removed=removed) <= risk            # support <= n_bar with the removal accounted for: certified
    assert md.scenario_risk(S1 - 1, n_bar, removed=removed) > risk         # ... and S1 is the smallest such sample size
    assert md.scenario_risk(S0, n_bar, removed=removed) > risk             # the sample size of the no-removal case does NOT certify it
    # the same compression-set size reached the other way round gives the same bound
    assert md.scenario_risk(S1, n_bar + removed, removed=0) == md.scenario_risk(S1, n_bar, removed=removed)
