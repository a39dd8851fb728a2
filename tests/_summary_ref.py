"""CPU checker for the device summaries: numpy restatement of misopy/credible_intervals.py:31-72
(sort + index) and of the device's fixed-order mean.  Test infrastructure only."""
import math

import numpy as np


def py2_round(x):
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def credible_interval(col, confidence_level=0.95):
    """credible_intervals.py:31-55 on one column of samples."""
    n = len(col)
    alpha = 1 - confidence_level
    lo = py2_round((alpha / 2) * n) - 1
    hi = py2_round((1 - alpha / 2) * n) - 1
    assert lo > 0 and hi > 0
    s = np.sort(col)
    return s[lo], s[hi]


def tree_mean(col):
    """The device's summation order: 256 strided partial sums, then a binary tree."""
    part = np.zeros(256)
    for t in range(256):
        acc = 0.0
        for v in col[t::256]:
            acc = acc + float(v)
        part[t] = acc
    stride = 128
    while stride >= 1:
        part[:stride] = part[:stride] + part[stride:2 * stride]
        stride //= 2
    return part[0] / float(len(col))


def summarize(samples_rows, confidence_level=0.95):
    """samples_rows: [K, S] (the C layout of `samples`).  -> mean[K], lo[K], hi[K]"""
    a = np.asarray(samples_rows, dtype=np.float64)
    m = np.array([tree_mean(r) for r in a])
    ci = [credible_interval(r, confidence_level) for r in a]
    return m, np.array([c[0] for c in ci]), np.array([c[1] for c in ci])
