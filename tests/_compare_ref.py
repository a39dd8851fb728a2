"""CPU checker for the device Bayes factors: numpy restatement of
misopy/hypothesis_test.py:89-179 (delta densities) and :348-380 (Bayes factor), with the Gaussian
KDE of scipy.stats.gaussian_kde (third-party, outside /root/reference; the reference subclasses it
with a constant covariance factor, hypothesis_test.py:41-59) written out:
    cov = var(delta, ddof=1) * factor^2;  f(0) = sum exp(-delta^2 / (2 cov)) / (n sqrt(2 pi cov)).
Test infrastructure only."""
import numpy as np

MAX_BF = 1e12


def kde_at_zero(delta, factor=0.3):
    n = len(delta)
    cov = np.cov(delta, rowvar=1, bias=False) * factor ** 2
    return float(np.sum(np.exp(-(delta * delta) / (2 * cov))) / (n * np.sqrt(2 * np.pi * cov)))


def bayes_factor(x1, x2, factor=0.3):
    """-> (bayes_factor, posterior density at 0) for one isoform column pair."""
    d = np.asarray(x1, dtype=np.float64) - np.asarray(x2, dtype=np.float64)
    mad = np.mean(np.abs(d))
    all_same = bool(np.all(d - d[0] == 0))
    if mad <= .009 or all_same:           # NullPeakedDensity -> inf at 0 -> BF 0
        return 0.0, np.inf
    post = kde_at_zero(d, factor)
    if post == 0:
        return MAX_BF, post
    return min(1.0 / post, MAX_BF), post
