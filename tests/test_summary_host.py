"""Host side of the summary row (f2): rank rule, formatting and file layout of
misopy/credible_intervals.py:4-72 and samples_utils.py:263-329."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "miso_amd"))
import summary as summ  # noqa: E402
from _summary_ref import credible_interval, tree_mean, py2_round  # noqa: E402


def test_rank_rule_is_python2_rounding():
    assert summ.credible_interval_ranks(5000) == (124, 4874)
    assert summ.credible_interval_ranks(30000) == (749, 29249)
    # 0.025 * 100 = 2.5: Python 2 rounds half away from zero -> 3 (Python 3 would give 2)
    assert summ.credible_interval_ranks(100) == (2, 97)
    assert summ.credible_interval_ranks(60, 0.95)[0] == 1
    for n in (61, 100, 999, 5000, 12345):
        for level in (0.95, 0.9, 0.5):
            a = 1 - level
            assert summ.credible_interval_ranks(n, level) == (py2_round((a / 2) * n) - 1,
                                                              py2_round((1 - a / 2) * n) - 1)


def test_checker_interval_and_mean():
    x = np.arange(1000, dtype=np.float64)[::-1].copy() / 1000.0
    assert credible_interval(x) == (0.024, 0.974)
    rng = np.random.default_rng(0)
    y = rng.random(5000)
    assert abs(tree_mean(y) - y.mean()) < 1e-15


def test_format_two_and_many_isoforms():
    f = summ.format_credible_intervals("ev", [0.123, 0.877], [0.051, 0.7], [0.249, 0.9])
    assert f == ["ev", "0.12", "0.05", "0.25"]
    f = summ.format_credible_intervals("ev", [0.5, 0.3, 0.2], [0.4, 0.2, 0.1], [0.6, 0.4, 0.3])
    assert f == ["ev", "0.50,0.30,0.20", "0.40,0.20,0.10", "0.60,0.40,0.30"]


def test_write_summary_layout(tmp_path):
    header = {"isoforms": "['A_B','A']", "counts": "(1,0):3,(1,1):5", "assigned_counts": "0:6,1:2",
              "chrom": "chr1", "strand": "+", "mRNA_starts": "1,1", "mRNA_ends": "9,9"}
    fn = tmp_path / "s.miso_summary"
    n = summ.write_summary(str(fn), [("ev", [0.1, 0.9], [0.05, 0.8], [0.2, 0.95], header),
                                     ("e2", [0.1, 0.9], [0.05, 0.8], [0.2, 0.95], {**header, "chrom": "NA"})])
    assert n == 2
    lines = fn.read_text().splitlines()
    assert lines[0] == "event_name\tmiso_posterior_mean\tci_low\tci_high\tisoforms\tcounts\t" \
                       "assigned_counts\tchrom\tstrand\tmRNA_starts\tmRNA_ends"
    assert lines[1] == "ev\t0.10\t0.05\t0.20\t['A_B','A']\t(1,0):3,(1,1):5\t0:6,1:2\tchr1\t+\t1,1\t9,9"
    assert lines[2].split("\t")[7] == "NA"
