"""GPU: the paired-end read loops over the dense records (kernels_grp.inl pe_dense, sampler_k2 MODE 2) against
the oracle's counter mode -- several chains per event (chains sharing a wavefront, the LDS score tables of
the 3-4 isoform class), events of a few pairs (padding reads, the quad of padding reads behind the last one),
and the cold exact scan (MISO_PE_FORCE_EXACT=1 sends every read there) against the fast path."""
import os

import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
from _problems import flat
from miso_amd import workload

pytestmark = pytest.mark.gpu
KW = dict(iters=300, burn=100, lag=2, chains=3)


def _batch(K, n_pairs, n_events, **extra):
    exons, isoforms, pos, cig = workload.event_reads(7, K, n_pairs, paired=True)
    b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, counts_trace=True, **dict(KW, **extra))
    for _ in range(n_events):
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    return b, (exons, isoforms, pos, cig)


@pytest.mark.parametrize("K,n_pairs,n_events", [(2, 3, 2), (2, 40, 7), (2, 401, 3), (2, 1000, 64), (3, 600, 5), (4, 37, 3),
                                                (5, 600, 5), (6, 250, 2), (10, 600, 5), (18, 333, 2), (20, 90, 3)])
def test_dense_records_against_the_oracle(orc, K, n_pairs, n_events):
    b, (exons, isoforms, pos, cig) = _batch(K, n_pairs, n_events)
    b.run(seed=5, first_event_id=0)
    g = orc.gene(flat(exons), isoforms)
    for e in range(min(n_events, 3)):
        cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=e, trace=True, **KW)
        assert cpu.rc == 0
        r = b.result(e, trace=True)
        where = (K, n_pairs, e, b.last_kernels())
        assert np.array_equal(r.counts_trace, cpu.trace["counts_trace"]), where
        assert np.array_equal(r.counts_hash, cpu.trace["counts_hash"]), where
        assert np.array_equal(r.samples, cpu.samples) and np.array_equal(r.loglik, cpu.loglik, equal_nan=True), where
        assert np.array_equal(r.assignment, cpu.assignment), where


@pytest.mark.parametrize("K,n_pairs,lanes", [(3, 1000, 8), (4, 1000, 8), (3, 600, 16), (4, 1000, 16), (5, 1000, 8), (8, 1000, 8),
                                             (9, 1000, 8)])
def test_eight_lanes_per_chain_and_score_tables_in_global_memory(orc, K, n_pairs, lanes, monkeypatch):
    """runtime.hip's rule for large batches of one isoform-count class -- eight lanes per chain, and for three and four
    isoforms the score tables in global memory instead of the LDS slice -- forced on a small batch (MISO_GENERAL_LANES;
    MISO_LDS_MAX_KB below what the tables need): the same bits as the checker."""
    monkeypatch.setenv("MISO_GENERAL_LANES", str(lanes))
    if K <= 4:
        monkeypatch.setenv("MISO_LDS_MAX_KB", "48" if lanes == 8 else "40")
    b, (exons, isoforms, pos, cig) = _batch(K, n_pairs, 5)
    b.run(seed=5, first_event_id=0)
    kc = 4 if K <= 4 else 8 if K <= 8 else 12
    assert "sampler_grp<%d, true, %d>" % (lanes, kc) in b.last_kernels(), b.last_kernels()
    g = orc.gene(flat(exons), isoforms)
    for e in range(3):
        cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=e, trace=True, **KW)
        assert cpu.rc == 0
        r = b.result(e, trace=True)
        where = (K, n_pairs, e, b.last_kernels())
        assert np.array_equal(r.counts_trace, cpu.trace["counts_trace"]), where
        assert np.array_equal(r.counts_hash, cpu.trace["counts_hash"]), where
        assert np.array_equal(r.samples, cpu.samples) and np.array_equal(r.loglik, cpu.loglik, equal_nan=True), where
        assert np.array_equal(r.assignment, cpu.assignment), where


@pytest.mark.parametrize("K", [3, 4, 7, 12, 18])
def test_exact_scan_equals_fast_path(K):
    """pe_dense's cold path (the reference's scan as written, taken when rnd < T does not hold) on every read."""
    def run():
        b, _ = _batch(K, 350, 4)
        b.run(seed=11, first_event_id=3)
        return [b.result(e, trace=True) for e in range(4)], b.last_kernels()
    old = os.environ.pop("MISO_PE_FORCE_EXACT", None)
    try:
        fast, kern = run()
        os.environ["MISO_PE_FORCE_EXACT"] = "1"
        exact, _ = run()
    finally:
        os.environ.pop("MISO_PE_FORCE_EXACT", None)
        if old is not None:
            os.environ["MISO_PE_FORCE_EXACT"] = old
    assert "sampler_grp" in kern
    for f, x in zip(fast, exact):
        assert np.array_equal(f.counts_trace, x.counts_trace) and np.array_equal(f.samples, x.samples)
        assert np.array_equal(f.loglik, x.loglik, equal_nan=True) and np.array_equal(f.assignment, x.assignment)


def test_plain_records_equal_dense_records():
    """MISO_NO_PE_DENSE=1 / MISO_NO_PE_DELTA=1: the round-1 loops over the plain records give the same results."""
    out = {}
    for tag, env in (("dense", {}), ("plain", {"MISO_NO_PE_DENSE": "1", "MISO_NO_PE_DELTA": "1"})):
        saved = {k: os.environ.pop(k, None) for k in ("MISO_NO_PE_DENSE", "MISO_NO_PE_DELTA")}
        os.environ.update(env)
        try:
            res = []
            for K in (2, 3, 9):
                b, _ = _batch(K, 500, 3)
                b.run(seed=2, first_event_id=0)
                res += [b.result(e, trace=True) for e in range(3)]
            out[tag] = res
        finally:
            for k in env:
                os.environ.pop(k, None)
            for k, v in saved.items():
                if v is not None:
                    os.environ[k] = v
    for d, p in zip(out["dense"], out["plain"]):
        assert np.array_equal(d.counts_trace, p.counts_trace) and np.array_equal(d.samples, p.samples)
        assert np.array_equal(d.loglik, p.loglik, equal_nan=True) and np.array_equal(d.assignment, p.assignment)


@pytest.mark.parametrize("K,sd", [(2, 60.0), (3, 60.0), (10, 45.0), (4, 31.0), (4, 32.0)])
def test_wide_fragment_distributions(orc, K, sd):
    """More than 254 fragment lengths (mean +- 4 sd with sd > 31): the byte records of pe_dense do not apply and
    the plain records' loops run; sampler_k2's u16 records still do.  sd = 31 / 32 straddle the limit."""
    mean, var = 300.0, sd * sd
    exons, isoforms, expr = workload.event_gene(11, K, min_len=500, max_len=900, gap=300)
    g = orc.gene(flat(exons), isoforms)
    orc.rng_seed(77)
    rc, iso, pos, cig = orc.simulate_paired_reads(g, [1.0 / K] * K, 500, 36, mean, var)
    assert rc == 0
    kw = dict(iters=200, burn=50, lag=2, chains=2)
    b = miso_amd.Batch(36, paired=True, mean=mean, var=var, counts_trace=True, **kw)
    for _ in range(3):
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=4, first_event_id=0)
    for e in range(3):
        cpu = orc.miso_paired(g, pos, cig, 36, mean, var, mode=OrcLib.COUNTER, seed=4, event_id=e, trace=True, **kw)
        assert cpu.rc == 0
        r = b.result(e, trace=True)
        where = (K, sd, e, b.last_kernels())
        assert np.array_equal(r.counts_trace, cpu.trace["counts_trace"]), where
        assert np.array_equal(r.samples, cpu.samples) and np.array_equal(r.loglik, cpu.loglik, equal_nan=True), where
        assert np.array_equal(r.assignment, cpu.assignment), where
