"""Host side of the comparison row (f3): the checker against scipy's gaussian_kde (the reference's
dependency), and the `.miso_bf` formatting of hypothesis_test.py:186-345."""
import os
import sys

import numpy as np
from scipy import stats

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "miso_amd"))
import compare as cmp  # noqa: E402
from _compare_ref import bayes_factor, kde_at_zero  # noqa: E402


class KdeCovfact(stats.gaussian_kde):
    """What hypothesis_test.py:41-59 builds: gaussian_kde with a constant covariance factor."""
    def __init__(self, dataset, covfact):
        self.covfact = covfact
        stats.gaussian_kde.__init__(self, dataset)

    def covariance_factor(self):
        return float(self.covfact)


def test_checker_equals_scipy_kde():
    rng = np.random.default_rng(0)
    for n, scale, shift in ((5000, 0.05, 0.0), (2700, 0.1, 0.2), (400, 0.02, -0.1), (5000, 0.03, 0.6)):
        d = rng.normal(shift, scale, n)
        want = KdeCovfact(d, 0.3).evaluate([0])[0]
        got = kde_at_zero(d, 0.3)
        assert abs(got - want) <= 1e-12 * max(want, 1e-300)


def test_checker_branches():
    rng = np.random.default_rng(1)
    a = rng.beta(20, 30, 3000)
    assert bayes_factor(a, a) == (0.0, np.inf)                       # all deltas identical
    assert bayes_factor(a, a + rng.normal(0, 1e-3, 3000))[0] == 0.0  # mean |delta| <= 0.009
    b = rng.beta(30, 20, 3000)
    bf, post = bayes_factor(a, b)
    assert 0 < bf < 1e12 and abs(bf * post - 1) < 1e-12
    far = bayes_factor(np.full(3000, 0.05) + rng.normal(0, 1e-3, 3000),
                       np.full(3000, 0.95) + rng.normal(0, 1e-3, 3000))
    assert far == (1e12, 0.0)                                        # density underflows -> cap


def test_two_isoform_fields():
    f = cmp.comparison_fields("ev", ([0.805, 0.195], [0.7, 0.1], [0.9, 0.3]),
                              ([0.2349999, 0.7650001], [0.15, 0.6], [0.33, 0.8]), [123.456, 123.456])
    # Decimal('0.805').quantize(0.01) is half-even -> 0.80; diff of the QUANTISED means
    assert f == ["ev", "0.80", "0.70", "0.90", "0.23", "0.15", "0.33", "0.57", "123.46"]


def test_multi_isoform_fields_and_file(tmp_path):
    s1 = ([0.5, 0.3, 0.2], [0.4, 0.2, 0.1], [0.6, 0.4, 0.3])
    s2 = ([0.25, 0.35, 0.4], [0.2, 0.3, 0.3], [0.3, 0.4, 0.5])
    f = cmp.comparison_fields("g", s1, s2, [5.0, 0.5, 1e12])
    assert f[1] == "0.50,0.30,0.20" and f[7] == "0.25,-0.05,-0.20"
    assert f[8] == "5.00,0.50,1000000000000.00"
    h = {"isoforms": "['a','b','c']", "counts": "(1,0,0):3", "assigned_counts": "0:3", "chrom": "chr9",
         "strand": "-", "mRNA_starts": "1,2,3", "mRNA_ends": "7,8,9"}
    h2 = dict(h, counts="(0,1,0):5", assigned_counts="1:5")
    fn = tmp_path / "a_vs_b.miso_bf"
    assert cmp.write_comparison(str(fn), [("g", s1, s2, [5.0, 0.5, 1e12], h, h2)]) == 1
    lines = fn.read_text().splitlines()
    assert lines[0].split("\t") == cmp.HEADER_FIELDS and len(lines) == 2
    cols = lines[1].split("\t")
    assert len(cols) == len(cmp.HEADER_FIELDS)
    assert cols[9:14] == ["['a','b','c']", "(1,0,0):3", "0:3", "(0,1,0):5", "1:5"]
    assert cols[14:] == ["chr9", "-", "1,2,3", "7,8,9"]
