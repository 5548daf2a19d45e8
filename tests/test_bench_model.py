"""bench.py's roofline model is SURVEY.md 8(d)'s: compulsory bytes and algorithmic flops per solve, checked against the
figures stated there (cfg 2: 23,976 B; ~1.06e7 flop at 10 RTI x 10 IPM iterations).  No GPU needed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def test_bytes_per_solve_is_the_survey_figure():
    assert bench.bytes_per_solve() == 8 * (2700 + 147 + 5 + 105 + 40) == 23976           # cfg 2
    assert bench.bytes_per_solve(npar=83) == 15656                                        # cfg 1
    assert bench.bytes_per_solve(npar=175) == 30376                                       # cfg 4


def test_flops_per_solve_is_the_survey_model():
    f = bench.flops_per_solve(10, 10)
    assert abs(f - 1.06e7) / 1.06e7 < 0.01
    # linear in the RTI count, affine in the IPM count
    assert bench.flops_per_solve(5, 10) * 2 == f
    assert bench.flops_per_solve(10, 4) < f


def test_usable_cpus_is_positive_and_bounded_by_the_visible_cpus():
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
