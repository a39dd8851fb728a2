"""CPU: the two-sample |delta psi| statistics (tests/_dpsi.py) that bench.py and the GPU tests use to compare the build's
sampler with the reference's (misopy/miso.c:845-900 against the kernels; summaries as credible_intervals.py:31-55) --
calibrated under the null, and with power against the shifts round 3's one-run 4 x MCSE rule could not see."""
import numpy as np

import _dpsi


def fake_runs(rng, S, E, K, shift=0.0, widen=1.0):
    """S runs of E events: per event a Dirichlet centre, per run correlated noise on the mean and both bounds (the cells of
    one run move together, the isoforms sum to one: everything the cell-wise binomial count of round 3 ignored)."""
    out = np.full((S, E, 4, K), np.nan)
    for e in range(E):
        centre = rng.dirichlet(np.ones(K))
        sd = 0.01 * (1 + e % 3)
        for s in range(S):
            d = rng.standard_normal(K) * sd
            d -= d.mean()                                   # sums to one
            m = centre + d + shift * sd * np.r_[1.0, -np.ones(K - 1) / (K - 1)]
            half = 0.05 * widen * (1 + 0.1 * rng.standard_normal())
            out[s, e, 0], out[s, e, 1], out[s, e, 2] = m, m - half, m + half
            out[s, e, 3] = out[s, e, 2] - out[s, e, 1]
    return out


def test_null_is_calibrated_and_a_one_mcse_bias_is_found():
    rng = np.random.default_rng(11)
    ps = []
    for rep in range(40):
        x = fake_runs(rng, 16, 24, 3)
        r = _dpsi.two_sample(x[:8], x[8:], n_perm=499, seed=rep, alpha=0.05)
        ps.append(r["p_row"])
        assert set(r["tests"]) == set(_dpsi.TESTS)
    ps = np.array(ps)
    assert (ps < 0.01).sum() <= 2 and np.median(ps) > 0.2     # Bonferroni over seven tests: conservative, never wild
    # group a shifted by ONE Monte-Carlo standard error in isoform 0 (round 3's rule -- every cell within 4 x MCSE -- says
    # "pass"): found by the signed tests
    x = fake_runs(np.random.default_rng(21), 16, 48, 3)
    y = x.copy()
    sd = np.array([0.01 * (1 + e % 3) for e in range(48)])
    y[:8, :, 0:3, 0] += sd[None, :, None]                    # + 1 MCSE on isoform 0's mean and bounds
    r = _dpsi.two_sample(y[:8], y[8:], n_perm=1999, alpha=0.01)
    assert not r["pass"] and r["tests"]["shift_mean"]["p"] <= 1e-3 and r["p_sign"] <= 1e-3
    # a fifth of a Monte-Carlo standard error is still visible over 48 events x 8 + 8 streams at the 1 % level
    y = x.copy()
    y[:8, :, 0:3, 0] += 0.35 * sd[None, :, None]
    r = _dpsi.two_sample(y[:8], y[8:], n_perm=1999, alpha=0.01)
    assert r["tests"]["shift_mean"]["p"] < 0.05
    # under-dispersed chains (intervals 15 % too narrow, means right): the width test
    y = x.copy()
    mid = 0.5 * (y[:8, :, 1] + y[:8, :, 2])
    y[:8, :, 1] = mid - 0.85 * (mid - y[:8, :, 1])
    y[:8, :, 2] = mid + 0.85 * (y[:8, :, 2] - mid)
    y[:8, :, 3] = y[:8, :, 2] - y[:8, :, 1]
    r = _dpsi.two_sample(y[:8], y[8:], n_perm=1999, alpha=0.01)
    assert not r["pass"] and r["tests"]["width"]["p"] <= 1e-3


def test_ragged_isoform_counts_and_flat_cells():
    """Whole-gene mixes: events of different isoform counts share one array (NaN beyond an event's K); a statistic pinned
    at a bound in every run of both groups (psi = 0 exactly) is no evidence either way, pinned at DIFFERENT values it is."""
    rng = np.random.default_rng(3)
    runs_a, runs_b = {}, {}
    for e, K in enumerate([3, 5, 8, 4]):
        c = rng.dirichlet(np.ones(K))
        for dst in (runs_a, runs_b):
            dst[e] = []
            for s in range(6):
                m = c + 0.01 * rng.standard_normal(K)
                lo, hi = m - 0.04, m + 0.04
                lo[0] = 0.0                                   # pinned in every run
                dst[e].append((m, lo, hi))
    a, b = _dpsi.stack_runs(runs_a, [0, 1, 2, 3], 8), _dpsi.stack_runs(runs_b, [0, 1, 2, 3], 8)
    assert a.shape == (6, 4, 4, 8) and np.isnan(a[0, 0, 0, 3:]).all() and not np.isnan(a[0, 2]).any()
    r = _dpsi.two_sample(a, b, n_perm=499, alpha=0.05)
    assert r["pass"] and np.isfinite(r["max_z"])
    b2 = b.copy()
    b2[:, :, 1, 0] = 0.01                                     # the other sampler pins that bound elsewhere
    b2[:, :, 3, 0] = b2[:, :, 2, 0] - b2[:, :, 1, 0]
    r = _dpsi.two_sample(a, b2, n_perm=1999, alpha=0.1)
    # (4 events x 6 + 6 runs: a relabelling separates one event's runs perfectly with probability 4 x 2 / 924)
    assert not r["pass"] and r["tests"]["max"]["p"] < 0.015 and r["tests"]["max"]["stat"] == 1e3
