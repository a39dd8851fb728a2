"""CPU: statistical parity of the counter-mode contract with the reference algorithm.

The reference draws from one sequential stream, the device contract from addressed Philox draws
and deterministic log/exp, so between them parity is distributional:
posterior mean and 95 % interval bounds of psi must agree within 4 x the Monte-Carlo standard
error estimated from 8 seeds.  An independent 1-D quadrature of the K = 2 posterior anchors both.

The four sampler outputs shipped with the reference (misopy/sashimi_plot/test-data/miso-data/*/
chr17/*.miso) are NOT usable as known answers for this path: they were written by a sampler whose
likelihood is not the C core's (no read length / overhang reproduces them through the C core's
model: best worst-case |delta psi| over read length 25..79, overhang 1..11 is 0.055), see
`test_sashimi_files_are_from_another_model`.
"""
import numpy as np
import pytest

from _libs import OrcLib
from _problems import flat, simulate_pe, simulate_se


def quad_posterior(counts, eff, n=100001):
    """K = 2: p(psi0 | reads) ~ prod_classes (sum_{k in class} psi_k)^n / (sum_j psi_j eff_j)^N."""
    x = np.linspace(0, 1, n)[1:-1]
    psi = np.stack([x, 1 - x])
    den = psi[0] * eff[0] + psi[1] * eff[1]
    logp = np.zeros_like(x)
    for mask, c in counts.items():
        if any(mask):
            logp += c * (np.log(sum(psi[k] for k in range(2) if mask[k])) - np.log(den))
    w = np.exp(logp - logp.max())
    w /= w.sum()
    cdf = np.cumsum(w)
    return (w * x).sum(), x[np.searchsorted(cdf, 0.025)], x[np.searchsorted(cdf, 0.975)]


def summaries(runs):
    """mean, 2.5 % and 97.5 % order statistics (credible_intervals.py:31-55) per run."""
    out = []
    for s in runs:
        v = np.sort(s[:, 0])
        n = len(v)
        out.append((v.mean(), v[int(round(0.025 * n)) - 1], v[int(round(0.975 * n)) - 1]))
    return np.array(out)


def test_k2_stream_counter_and_quadrature_agree(orc):
    exons, isoforms, g, pos, cig = simulate_se(orc, 2, 1000, seed=42)
    probe = orc.miso(g, pos, cig, 36, iters=20, burn=2, lag=1, chains=1)
    counts = {tuple(int(v) for v in t): c for t, c in zip(probe.class_templates, probe.class_counts)}
    eff = [n - 36 + 1 for n in orc.isolength(g)]
    q = np.array(quad_posterior(counts, eff))
    stream, counter = [], []
    for s in range(8):
        orc.rng_seed(1000 + s)
        stream.append(orc.miso(g, pos, cig, 36, iters=4000, burn=1000, lag=1, chains=1).samples)
        counter.append(orc.miso(g, pos, cig, 36, iters=4000, burn=1000, lag=1, chains=1,
                                mode=OrcLib.COUNTER, seed=500 + s, event_id=s).samples)
    S, Cn = summaries(stream), summaries(counter)
    mcse = np.sqrt(S.var(0, ddof=1) / 8 + Cn.var(0, ddof=1) / 8)
    assert (np.abs(S.mean(0) - Cn.mean(0)) < 4 * mcse).all(), (S.mean(0), Cn.mean(0), mcse)
    for name, X in (("stream", S), ("counter", Cn)):
        se = np.sqrt(X.var(0, ddof=1) / 8)
        assert (np.abs(X.mean(0) - q) < 4 * se + 2e-3).all(), (name, X.mean(0), q, se)


@pytest.mark.parametrize("K", [3, 5])
def test_multi_isoform_stream_vs_counter(orc, K):
    exons, isoforms, g, pos, cig = simulate_se(orc, K, 800, seed=60 + K)
    a, b = [], []
    for s in range(8):
        orc.rng_seed(2000 + s)
        a.append(orc.miso(g, pos, cig, 36, iters=3000, burn=1000, lag=2, chains=1).samples.mean(0))
        b.append(orc.miso(g, pos, cig, 36, iters=3000, burn=1000, lag=2, chains=1,
                          mode=OrcLib.COUNTER, seed=77 + s, event_id=3).samples.mean(0))
    a, b = np.array(a), np.array(b)
    mcse = np.sqrt(a.var(0, ddof=1) / 8 + b.var(0, ddof=1) / 8)
    assert (np.abs(a.mean(0) - b.mean(0)) < 4 * mcse + 1e-3).all(), (a.mean(0), b.mean(0), mcse)


def test_paired_end_stream_vs_counter(orc):
    exons, isoforms, g, pos, cig = simulate_pe(orc, 2, 400, seed=70)
    a, b = [], []
    for s in range(8):
        orc.rng_seed(3000 + s)
        a.append(orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, iters=2500, burn=500, lag=2,
                                 chains=1).samples[:, 0].mean())
        b.append(orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, iters=2500, burn=500, lag=2, chains=1,
                                 mode=OrcLib.COUNTER, seed=9 + s, event_id=1).samples[:, 0].mean())
    mcse = np.sqrt(np.var(a, ddof=1) / 8 + np.var(b, ddof=1) / 8)
    assert abs(np.mean(a) - np.mean(b)) < 4 * mcse + 1e-3, (np.mean(a), np.mean(b), mcse)


@pytest.mark.parametrize("K,n_pairs", [(3, 500), (5, 600), (10, 700)])
def test_paired_end_multi_isoform_stream_vs_counter_and_reference(orc, ref, K, n_pairs):
    """The paired-end contract changes the arithmetic of the read score (2^-26 fixed point, order-free, where
    miso_paired.c:133-174, 393-419 sums doubles read by read) as well as the random stream: three or more isoforms,
    ALL isoforms, posterior means AND both Chen-Shao bounds (credible_intervals.py:31-55), 8 seeds each way --
    the oracle's stream mode, its counter mode (= the GPU bit for bit) and the REAL reference must agree within
    4 x MCSE.  (Two isoforms: test_paired_end_stream_vs_counter.)"""
    exons, isoforms, g, pos, cig = simulate_pe(orc, K, n_pairs, seed=170 + K)
    gr = ref.gene(flat(exons), isoforms)

    def stats(samples):
        x = np.sort(samples, axis=0)
        n = len(x)
        return np.concatenate([x.mean(0), x[int(round(0.025 * n)) - 1], x[int(round(0.975 * n)) - 1]])
    kw = dict(iters=3000, burn=1000, lag=2, chains=1)
    a, b, c = [], [], []
    for s in range(8):
        orc.rng_seed(5000 + s)
        a.append(stats(orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, **kw).samples))
        b.append(stats(orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=31 + s, event_id=4, **kw).samples))
        ref.rng_seed(6000 + s)
        c.append(stats(ref.miso_paired(gr, pos, cig, 36, 250.0, 900.0, **kw).samples))
    a, b, c = np.array(a), np.array(b), np.array(c)
    for x, y, what in ((a, b, "stream vs counter"), (c, b, "reference vs counter")):
        mcse = np.sqrt(x.var(0, ddof=1) / 8 + y.var(0, ddof=1) / 8)
        d = np.abs(x.mean(0) - y.mean(0))
        # 3 K statistics, P(|t_14| > 4) ~ 1e-3 each: at most one just beyond 4 x MCSE (+ 1e-3 absolute), none beyond 6.5
        assert (d > 4 * mcse + 1e-3).sum() <= 1 and (d <= 6.5 * mcse + 1e-3).all(), (what, K, d / np.maximum(mcse, 1e-12))


def test_count_sums_equal_per_read_sums_up_to_rounding(orc):
    """The contract sums scores from per-isoform counts; the reference sums per read. Same
    assignments -> same psi trajectory, log scores equal to ~1e-10."""
    exons, isoforms, g, pos, cig = simulate_se(orc, 4, 600, seed=81)
    kw = dict(iters=800, burn=100, lag=1, chains=2, mode=OrcLib.COUNTER, seed=5, event_id=2)
    a = orc.miso(g, pos, cig, 36, **kw)
    b = orc.miso(g, pos, cig, 36, per_read_sums=True, **kw)
    assert np.array_equal(a.samples, b.samples) and np.array_equal(a.assignment, b.assignment)
    assert np.abs(a.loglik - b.loglik).max() < 1e-9 * np.abs(a.loglik).max()


def test_sashimi_files_are_from_another_model(orc):
    """Class counts from the four shipped .miso headers; file means 0.789/0.759/0.248/0.249."""
    cases = [((1, 21, 23), 0.789), ((7, 54, 63), 0.759), ((11, 5, 17), 0.248), ((12, 7, 16), 0.249)]
    worst_best = 1.0
    for rl in range(25, 80, 3):
        for ov in range(1, 12, 2):
            eff = [210 - rl + 1 - 4 * (ov - 1), 171 - rl + 1 - 2 * (ov - 1)]
            if ov >= rl / 2 or min(eff) <= 0:
                continue
            err = max(abs(quad_posterior({(0, 1): c[0], (1, 0): c[1], (1, 1): c[2]}, eff, 4001)[0] - t)
                      for c, t in cases)
            worst_best = min(worst_best, err)
    assert worst_best > 0.04
    # ... while the C-core model (= oracle) and the quadrature agree on such small events too
    exons, isoforms = [(1, 91), (201, 239), (401, 480)], [[0, 1, 2], [0, 2]]
    g = orc.gene(flat(exons), isoforms)
    reads = [(74, b"18M309N14M")] * 11 + [(205, b"32M")] * 5 + [(10, b"32M")] * 17
    pos = np.array([p for p, _ in reads], np.int32)
    cig = [c for _, c in reads]
    means = []
    for s in range(6):
        orc.rng_seed(4000 + s)
        r = orc.miso(g, pos, cig, 32, iters=6000, burn=1000, lag=1, chains=1)
        means.append(r.samples[:, 0].mean())
    assert list(r.class_counts) == [11, 5, 17]
    q = quad_posterior({(0, 1): 11, (1, 0): 5, (1, 1): 17}, [210 - 31, 171 - 31])[0]
    assert abs(np.mean(means) - q) < 4 * np.std(means, ddof=1) / np.sqrt(6) + 2e-3
