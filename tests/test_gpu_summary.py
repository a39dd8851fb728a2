"""Device-side posterior summaries against the CPU restatement of
misopy/credible_intervals.py; bit-exact (the interval bounds ARE samples; the mean uses a fixed
summation order the checker reproduces)."""
import os
import sys

import numpy as np
import pytest

from _problems import se_gene, expr_for
from _summary_ref import summarize, credible_interval
from miso_amd import capi

pytestmark = pytest.mark.gpu


def se_problem(K, n_reads, seed):
    exons, isoforms = se_gene(K)
    return dict(gene=capi.Gene(exons, isoforms), expression=expr_for(K), n_reads=n_reads, sim_seed=seed)


def pe_problem(K, n_reads, seed):
    exons, isoforms = se_gene(K, exlen=500, gap=300)
    return dict(gene=capi.Gene(exons, isoforms), expression=expr_for(K), n_reads=n_reads, sim_seed=seed)


def _fill(b, problems):
    for p in problems:
        b.add_simulated(p["gene"], p["expression"], p["n_reads"], p["sim_seed"])
    b._keep = problems
    return b


def _batch(problems, paired=False, **kw):
    if paired:
        kw.update(mean=250.0, var=900.0)
    b = _fill(capi.Batch(36, paired=paired, **kw), problems)
    b.run(seed=7)
    return b


def _cols(r):
    return r.samples.T          # [K, S]


@pytest.mark.parametrize("K", [2, 3, 5, 10])
def test_summary_matches_checker_se(K):
    probs = [se_problem(K=K, n_reads=150 + 37 * i, seed=100 + i) for i in range(5)]
    b = _batch(probs, iters=1200, burn=200, lag=2, chains=3)
    b.summarize(0.95)
    for i in range(len(probs)):
        r = b.result(i)
        m, lo, hi = b.summary(i)
        em, elo, ehi = summarize(_cols(r), 0.95)
        assert np.array_equal(lo, elo) and np.array_equal(hi, ehi)
        assert np.array_equal(m, em)
        assert np.allclose(m, r.samples.mean(axis=0), rtol=0, atol=1e-13)
        assert abs(m.sum() - 1) < 1e-9


def test_summary_mixed_batch_and_levels():
    probs = [se_problem(K=2, n_reads=300, seed=1), se_problem(K=7, n_reads=90, seed=2),
             se_problem(K=2, n_reads=10, seed=3), se_problem(K=4, n_reads=500, seed=4)]
    b = _batch(probs, iters=2100, burn=100, lag=1, chains=2)
    for level in (0.95, 0.9, 0.5):
        b.summarize(level)
        for i in range(len(probs)):
            r = b.result(i)
            m, lo, hi = b.summary(i)
            em, elo, ehi = summarize(_cols(r), level)
            assert np.array_equal(lo, elo) and np.array_equal(hi, ehi) and np.array_equal(m, em)
            assert np.all(lo <= hi)


def test_summary_paired_and_trailing_zero_columns():
    # lag does not divide M - B: the trailing sample columns stay zero (quirk C8) and, as in the
    # reference's .miso files, they take part in the summary.
    probs = [pe_problem(K=2, n_reads=120, seed=5), pe_problem(K=3, n_reads=200, seed=6)]
    b = _batch(probs, paired=True, iters=1103, burn=100, lag=7, chains=2)
    b.summarize(0.95)
    for i in range(len(probs)):
        r = b.result(i)
        m, lo, hi = b.summary(i)
        em, elo, ehi = summarize(_cols(r), 0.95)
        assert np.array_equal(lo, elo) and np.array_equal(hi, ehi) and np.array_equal(m, em)


def test_summary_without_download_and_errors():
    b = _fill(capi.Batch(36, iters=1100, burn=100, lag=1, chains=1), [se_problem(K=2, n_reads=100, seed=9)])
    with pytest.raises(capi.InternalError):
        b.summarize(0.95)                      # nothing launched yet
    b.upload(0); b.launch(seed=3); b.sync()
    b.summarize(0.95)                          # no download in between
    m, lo, hi = b.summary(0)
    b.download()
    r = b.result(0)
    assert (lo[0], hi[0]) == credible_interval(r.samples[:, 0], 0.95)
    # too few samples for an interval: the reference asserts (credible_intervals.py:49-50)
    b2 = _fill(capi.Batch(36, iters=30, burn=10, lag=1, chains=1), [se_problem(K=2, n_reads=50, seed=9)])
    b2.run(seed=1)
    with pytest.raises(capi.InternalError, match="Too few samples"):
        b2.summarize(0.95)


def test_summary_file_from_batch(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "miso_amd"))
    import miso_sampler as ms
    import summary as summ
    exons = [(100, 199), (300, 359), (500, 599)]
    gene2 = ms.SimpleGene(exons, [[0, 1, 2], [0, 2]], label="ev2", chrom="chr1", strand="+")
    gene3 = ms.SimpleGene(exons + [(700, 799)], [[0, 1, 2, 3], [0, 2, 3], [0, 3]], label="ev3",
                          chrom="chr2", strand="-")
    rng = np.random.default_rng(0)
    def reads(n, hi):
        return [int(x) for x in rng.integers(99, hi, n)], ["36M"] * n
    params = ms.get_single_end_sampler_params(2, 36)
    s = ms.MISOSampler(params)
    events = [(reads(200, 160), gene2, str(tmp_path / "chr1" / "ev2")),
              (reads(150, 160), gene3, str(tmp_path / "chr2" / "ev3"))]
    out = s.run_sampler_batch(2000, events, num_chains=2, burn_in=200, lag=2, seed=11,
                              summary_file=str(tmp_path / "summary.miso_summary"))
    assert all(out)
    lines = open(tmp_path / "summary.miso_summary").read().splitlines()
    assert lines[0].split("\t") == summ.HEADER_FIELDS
    assert len(lines) == 3
    for line, fn in zip(lines[1:], out):
        f = line.split("\t")
        assert len(f) == len(summ.HEADER_FIELDS)
        samples, header, _ = ms.load_samples(fn)              # the 4-decimal text the reference reads
        K = samples.shape[1]
        exp_mean = samples.mean(axis=0)
        got_mean = [float(x) for x in f[1].split(",")]
        assert len(got_mean) == (K if K > 2 else 1)
        for k, g in enumerate(got_mean):
            assert abs(g - exp_mean[k]) <= 0.005 + 1e-4
            lo, hi = credible_interval(samples[:, k].copy(), 0.95)
            assert abs(float(f[2].split(",")[k]) - lo) <= 0.005 + 1e-4
            assert abs(float(f[3].split(",")[k]) - hi) <= 0.005 + 1e-4
        assert f[4] == header["isoforms"] and f[5] == header["counts"]
        assert f[6] == header["assigned_counts"] and f[7] == header["chrom"]


def test_summary_more_samples_than_the_register_cache():
    probs = [se_problem(K=2, n_reads=80, seed=21), se_problem(K=3, n_reads=60, seed=22)]
    b = _batch(probs, iters=3200, burn=200, lag=1, chains=3)      # S = 9000 > 8192
    b.summarize(0.95)
    for i in range(len(probs)):
        r = b.result(i)
        assert r.samples.shape[0] == 9000
        m, lo, hi = b.summary(i)
        em, elo, ehi = summarize(_cols(r), 0.95)
        assert np.array_equal(lo, elo) and np.array_equal(hi, ehi) and np.array_equal(m, em)


@pytest.mark.parametrize("K,paired,S_big", [(2, False, False), (5, False, False), (3, True, False), (2, False, True)])
def test_summary_of_the_file_text_equals_what_summarize_miso_reads(K, paired, S_big):
    """miso_batch_summarize_as_text: the reference summarises the `.miso` FILE (samples_utils.py:130-262 ->
    credible_intervals.py:4-72), i.e. every sample after "%.4f" and float().  The device rounds each sample to four
    decimals exactly as Python's % does and reads it back as float() would: the interval bounds equal the numpy
    restatement of credible_intervals.py on the parsed text bit for bit, the mean is the exact sum of the printed
    digits over n (numpy's float mean of the parsed values agrees to 1e-15)."""
    probs = [(pe_problem if paired else se_problem)(K=K, n_reads=120 + 61 * i, seed=300 + i) for i in range(4)]
    kw = dict(iters=9000, burn=500, lag=1, chains=1) if S_big else dict(iters=1500, burn=300, lag=2, chains=3)
    b = _batch(probs, paired=paired, **kw)
    b.summarize(0.95, as_text=True)
    for i in range(len(probs)):
        r = b.result(i)
        text = np.array([[float("%.4f" % v) for v in row] for row in r.samples])      # what the file hands on
        digits = np.array([[int(round(float("%.4f" % v) * 10000)) for v in row] for row in r.samples])
        m, lo, hi = b.summary(i)
        for k in range(K):
            elo, ehi = credible_interval(text[:, k], 0.95)
            assert lo[k] == elo and hi[k] == ehi, (i, k)
            assert m[k] == digits[:, k].sum() / (len(text) * 10000.0), (i, k)
            assert abs(m[k] - text[:, k].mean()) < 1e-15
    # the full-precision summary is a different thing (otherwise this test would show nothing)
    text_lo = [b.summary(i)[1].copy() for i in range(len(probs))]
    b.summarize(0.95)
    assert any(not np.array_equal(b.summary(i)[1], text_lo[i]) for i in range(len(probs)))
