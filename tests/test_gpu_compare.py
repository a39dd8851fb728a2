"""Device-side two-sample comparison against the numpy/scipy restatement of
misopy/hypothesis_test.py.  Means are bit-exact (fixed summation order); the Bayes factor is
floating point: tolerance 1e-9 relative (different summation order of S <= 10^4 terms and a
1-ulp exp)."""
import os
import sys

import numpy as np
import pytest

from _problems import se_gene, expr_for
from _compare_ref import bayes_factor
from _summary_ref import tree_mean
from miso_amd import capi

pytestmark = pytest.mark.gpu
BF_RTOL = 1e-9


def _sample(K, spec, seed, expr=None, paired=False, **kw):
    """spec: list of (n_reads, sim_seed, expression or None)."""
    exons, isoforms = se_gene(K, exlen=500, gap=300) if paired else se_gene(K)
    g = capi.Gene(exons, isoforms)
    if paired:
        kw.update(mean=250.0, var=900.0)
    b = capi.Batch(36, paired=paired, **kw)
    for n, sim_seed, ex in spec:
        b.add_simulated(g, expr_for(K) if ex is None else ex, n, sim_seed)
    b._gene = g
    b.run(seed=seed)
    return b


def _check(b1, b2, n, factor=0.3):
    b1.compare(b2, factor)
    for i in range(n):
        s1, s2 = b1.result(i).samples, b2.result(i).samples
        m1, m2, bf, dens = b1.comparison(i)
        for k in range(s1.shape[1]):
            assert m1[k] == tree_mean(s1[:, k]) and m2[k] == tree_mean(s2[:, k])
            ebf, edens = bayes_factor(s1[:, k], s2[:, k], factor)
            if ebf in (0.0, 1e12):
                assert bf[k] == ebf, (i, k, bf[k], ebf)
            else:
                assert abs(bf[k] - ebf) <= BF_RTOL * ebf, (i, k, bf[k], ebf)
                assert abs(dens[k] - edens) <= BF_RTOL * edens


@pytest.mark.parametrize("K", [2, 3, 6])
def test_bayes_factor_matches_checker(K):
    kw = dict(iters=2200, burn=200, lag=2, chains=3)
    e2 = np.arange(K, 0, -1, dtype=np.float64); e2 /= e2.sum()
    spec1 = [(200 + 50 * i, 10 + i, None) for i in range(6)]
    spec2 = [(200 + 50 * i, 90 + i, e2 if i % 2 else None) for i in range(6)]
    b1, b2 = _sample(K, spec1, seed=3, **kw), _sample(K, spec2, seed=4, **kw)
    _check(b1, b2, 6)
    _check(b1, b2, 6, factor=0.5)


def test_branches_null_peaked_and_cap():
    kw = dict(iters=1500, burn=500, lag=1, chains=1)
    far1 = np.array([0.97, 0.03]); far2 = np.array([0.03, 0.97])
    spec1 = [(300, 5, None), (4000, 6, far1), (30000, 8, None)]
    spec2 = [(300, 5, None), (4000, 7, far2), (30000, 8, None)]
    # event 0 + same seed: identical chains -> every delta identical -> BF 0
    b1, b2 = _sample(2, spec1, seed=11, **kw), _sample(2, spec2, seed=11, **kw)
    _check(b1, b2, 3)
    _, _, bf, dens = b1.comparison(0)
    assert bf[0] == 0.0 and np.isinf(dens[0])
    _, _, bf, _ = b1.comparison(1)
    assert bf[0] == 1e12                      # posteriors far apart: density underflows, capped
    # different seeds, same (many) reads: |delta| small -> the mean|delta| <= 0.009 branch
    b3 = _sample(2, spec2[2:], seed=12, **kw)
    b4 = _sample(2, spec1[2:], seed=13, **kw)
    _check(b4, b3, 1)
    assert b4.comparison(0)[2][0] == 0.0


def test_paired_end_and_errors():
    kw = dict(iters=1300, burn=300, lag=1, chains=2)
    b1 = _sample(3, [(200, 1, None), (150, 2, None)], seed=1, paired=True, **kw)
    b2 = _sample(3, [(220, 3, None), (180, 4, None)], seed=2, paired=True, **kw)
    _check(b1, b2, 2)
    b3 = _sample(3, [(220, 3, None)], seed=2, paired=True, **kw)
    with pytest.raises(capi.InternalError, match="differ"):
        b1.compare(b3)
    b5 = _sample(2, [(220, 3, None), (180, 4, None)], seed=2, **kw)
    with pytest.raises(capi.InternalError):
        b1.compare(b5)
    with pytest.raises(capi.InternalError, match="has not run"):
        b5.comparison(0)


def test_module_compare_batch_and_bf_file(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "miso_amd"))
    import miso_sampler as ms
    import compare as cmp
    exons = [(100, 199), (300, 359), (500, 599)]
    gene2 = ms.SimpleGene(exons, [[0, 1, 2], [0, 2]], label="ev2", chrom="chr1", strand="+")
    gene3 = ms.SimpleGene(exons + [(700, 799)], [[0, 1, 2, 3], [0, 2, 3], [0, 3]], label="ev3",
                          chrom="chr2", strand="-")
    rng = np.random.default_rng(0)

    def reads(n, lo, hi):
        return [int(x) for x in rng.integers(lo, hi, n)], ["36M"] * n
    s = ms.MISOSampler(ms.get_single_end_sampler_params(2, 36))
    ev1 = [(reads(300, 99, 160), gene2, str(tmp_path / "s1" / "chr1" / "ev2")),
           (reads(200, 99, 160), gene3, str(tmp_path / "s1" / "chr2" / "ev3"))]
    ev2 = [(reads(300, 280, 330), gene2, str(tmp_path / "s2" / "chr1" / "ev2")),
           (reads(200, 99, 160), gene3, str(tmp_path / "s2" / "chr2" / "ev3"))]
    bf_file = str(tmp_path / "s1_vs_s2.miso_bf")
    out = s.run_comparison_batch(2000, ev1, ev2, bf_file, num_chains=2, burn_in=200, lag=2, seed=5)
    assert all(a and b for a, b in out)
    lines = open(bf_file).read().splitlines()
    assert lines[0].split("\t") == cmp.HEADER_FIELDS and len(lines) == 3
    for line, (f1, f2) in zip(lines[1:], out):
        c = line.split("\t")
        assert len(c) == len(cmp.HEADER_FIELDS)
        x1, h1, _ = ms.load_samples(f1)
        x2, h2, _ = ms.load_samples(f2)
        K = x1.shape[1]
        got = [float(v) for v in c[8].split(",")]
        assert len(got) == (K if K > 2 else 1)
        for k, g in enumerate(got):                     # BF from the 4-decimal text the reference reads
            ebf, _ = bayes_factor(x1[:, k], x2[:, k])
            assert abs(g - ebf) <= 0.02 * max(ebf, 1.0) + 0.006
        assert c[10] == h1["counts"] and c[12] == h2["counts"] and c[14] == h1["chrom"]
