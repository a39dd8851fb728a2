"""CPU: the collapsed Gibbs step (include/miso_binomial.h, oracle ORC_MODE_COLLAPSED).

The reference reassigns every read by itself (miso.c:30-91 inside miso.c:493-552) and then uses the per-isoform
counts only (miso.c:243-307); single-end reads compatible with the same isoforms are exchangeable, so the counts of
such a class can be drawn directly -- a chain of exact binomials -- without changing the Markov chain on (psi, counts).
Checked here: (1) the binomial sampler against the exact probabilities (chi-square) in every regime of the
algorithm; (2) the collapsed chain against the per-read chain, the 1-D quadrature of the posterior and the REAL
reference (oracle/_ref, when built): posterior mean and both interval bounds within 4 Monte-Carlo standard errors."""
import numpy as np
import pytest
from scipy import stats

from _libs import OrcLib
from _problems import flat, simulate_se
from test_statistics import quad_posterior, summaries


@pytest.mark.parametrize("n,p", [(10, 0.3), (688, 0.31), (688, 0.5), (688, 0.93), (100000, 0.4), (50, 0.999),
                                  (1000, 0.0099), (1000, 0.0101), (3, 0.5), (21, 0.5), (20, 0.5), (100000, 0.00011),
                                  (1, 0.7), (60000, 0.97)])
def test_binomial_sampler_matches_the_exact_probabilities(orc, n, p):
    """inversion (n min(p, q) < 10) and BTRS on both sides of the switch, p on both sides of 1/2, tiny and huge n"""
    cnt = 300000
    x = orc.binomial(n, p, cnt, seed=11, event_id=5)
    assert x.min() >= 0 and x.max() <= n
    lo, hi = int(stats.binom.ppf(1e-7, n, p)), int(stats.binom.ppf(1 - 1e-7, n, p))
    ks = np.arange(lo, hi + 1)
    obs = np.bincount(np.clip(x, lo, hi) - lo, minlength=len(ks)).astype(float)
    pm = stats.binom.pmf(ks, n, p)
    pm[0] += stats.binom.cdf(lo - 1, n, p); pm[-1] += stats.binom.sf(hi, n, p)
    exp = pm * cnt
    keep = exp >= 10
    o, e = np.append(obs[keep], obs[~keep].sum()), np.append(exp[keep], exp[~keep].sum())
    if e[-1] < 1e-9:
        o, e = o[:-1], e[:-1]
    chi = ((o - e) ** 2 / np.maximum(e, 1e-300)).sum()
    assert stats.chi2.sf(chi, max(len(o) - 1, 1)) > 1e-4, (n, p, chi, len(o))
    assert abs(x.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / cnt) + 1e-9


def test_binomial_edge_cases(orc):
    assert (orc.binomial(50, 0.0, 100) == 0).all() and (orc.binomial(50, 1.0, 100) == 50).all()
    assert (orc.binomial(50, float("nan"), 100) == 0).all() and (orc.binomial(0, 0.5, 100) == 0).all()
    assert (orc.binomial(50, -0.1, 100) == 0).all() and (orc.binomial(50, 1.5, 100) == 50).all()


def test_k2_collapsed_per_read_and_quadrature_agree(orc):
    exons, isoforms, g, pos, cig = simulate_se(orc, 2, 1000, seed=42)
    probe = orc.miso(g, pos, cig, 36, iters=20, burn=2, lag=1, chains=1)
    counts = {tuple(int(v) for v in t): c for t, c in zip(probe.class_templates, probe.class_counts)}
    eff = [n - 36 + 1 for n in orc.isolength(g)]
    q = np.array(quad_posterior(counts, eff))
    per_read, coll = [], []
    for s in range(8):
        kw = dict(iters=4000, burn=1000, lag=1, chains=1, seed=500 + s, event_id=s)
        per_read.append(orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, **kw).samples)
        coll.append(orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, **kw).samples)
    P, Cn = summaries(per_read), summaries(coll)
    assert not np.array_equal(P, Cn)                         # different draws ...
    mcse = np.sqrt(P.var(0, ddof=1) / 8 + Cn.var(0, ddof=1) / 8)
    assert (np.abs(P.mean(0) - Cn.mean(0)) < 4 * mcse).all(), (P.mean(0), Cn.mean(0), mcse)   # ... one distribution
    se = np.sqrt(Cn.var(0, ddof=1) / 8)
    assert (np.abs(Cn.mean(0) - q) < 4 * se + 2e-3).all(), (Cn.mean(0), q, se)


@pytest.mark.parametrize("K,n", [(2, 40), (3, 800), (5, 800)])
def test_collapsed_against_the_real_reference(orc, ref, K, n):
    """the reference's own C core (stream RNG) against the collapsed chain, every isoform, mean and both bounds"""
    exons, isoforms, g, pos, cig = simulate_se(orc, K, n, seed=90 + K)
    rg = ref.gene(flat(exons), isoforms)
    a, b = [], []
    for s in range(8):
        ref.rng_seed(3000 + s)
        r = ref.miso(rg, pos, cig, 36, iters=3000, burn=1000, lag=2, chains=1)
        a.append(r.samples.reshape(-1, K))
        b.append(orc.miso(g, pos, cig, 36, iters=3000, burn=1000, lag=2, chains=1, mode=OrcLib.COLLAPSED,
                          seed=177 + s, event_id=3).samples.reshape(-1, K))

    def stats_of(runs):
        out = []
        for sm in runs:
            v = np.sort(sm, axis=0)
            m = len(v)
            out.append(np.concatenate([sm.mean(0), v[int(round(0.025 * m)) - 1], v[int(round(0.975 * m)) - 1]]))
        return np.array(out)
    A, B = stats_of(a), stats_of(b)
    mcse = np.sqrt(A.var(0, ddof=1) / 8 + B.var(0, ddof=1) / 8)
    assert (np.abs(A.mean(0) - B.mean(0)) < 4 * mcse + 1e-3).all(), (A.mean(0), B.mean(0), mcse)
