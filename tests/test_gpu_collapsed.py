"""The collapsed Gibbs step (csrc/kernels_lane.hip, include/miso_binomial.h): single-end two-isoform events draw the
COUNT of their exchangeable reads on isoform 0 as one exact binomial per iteration instead of one uniform per read
(reference: miso.c:30-91 per read, then counts only, miso.c:243-307).  Bit for bit against the checker's collapsed
mode (oracle/miso_oracle.c ORC_MODE_COLLAPSED) -- samples, log scores, per-iteration counts, the final per-read
assignment -- on events from 0 to 60 000 reads in one launch; the binomial sampler itself and the agreement of the
collapsed chain with the per-read chain and the real reference are CPU tests (tests/test_collapsed.py)."""
import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
from _problems import flat, se_gene, expr_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return OrcLib()


def _events(orc, sizes):
    evs = []
    for j, n in enumerate(sizes):
        exons, isoforms = se_gene(2, exlen=300 + 17 * j)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(4000 + j)
        rc, _, pos, cig = orc.simulate_reads(g, expr_for(2) if j % 3 else [0.03, 0.97], max(n, 1), 36)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:n], cig[:n]))
    return evs


@pytest.mark.parametrize("chains,iters,burn,lag", [(1, 300, 50, 1), (3, 200, 20, 4), (2, 0, 0, 1), (1, 1, 0, 1)])
def test_collapsed_bit_exact_against_the_checker(orc, chains, iters, burn, lag):
    sizes = [700, 60000, 20, 0, 3, 150, 9000, 45, 1000, 64, 65, 2500, 31, 1, 333, 5000]
    evs = _events(orc, sizes)
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains)
    b = miso_amd.Batch(36, counts_trace=True, collapsed=True, **kw)
    for exons, isoforms, g, pos, cig in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=77, first_event_id=1200)
    assert b.last_kernels() == "sampler_lane"
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, seed=77, event_id=1200 + i, trace=True, **kw)
        assert r.rc == 0
        gpu = b.result(i, trace=True)
        where = (sizes[i], kw)
        assert np.array_equal(gpu.counts_trace, r.trace["counts_trace"]), where
        assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
        assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
        assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
        assert (gpu.assignment == r.assignment).all(), where
        assert gpu.rundata.noAccepted == r.accepted, where


def test_collapsed_differs_from_per_read_draws_but_not_in_distribution(orc):
    """Same events through both modes: different draws, the same posterior (means within 4 Monte-Carlo standard errors
    estimated from 16 seeds of each)."""
    sizes = [700, 40, 2000, 150]
    evs = _events(orc, sizes)
    kw = dict(iters=3000, burn=500, lag=1, chains=1)
    means = {False: [], True: []}
    for collapsed in (False, True):
        for seed in range(16):
            b = miso_amd.Batch(36, collapsed=collapsed, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=500 + seed, first_event_id=0)
            means[collapsed].append([b.result(i).samples[:, 0].mean() for i in range(len(evs))])
    a, c = np.array(means[False]), np.array(means[True])
    assert not np.array_equal(a, c)
    se = np.sqrt(a.var(0, ddof=1) / len(a) + c.var(0, ddof=1) / len(c))
    assert (np.abs(a.mean(0) - c.mean(0)) < 4 * se + 1e-12).all(), (a.mean(0), c.mean(0), se)


def test_collapsed_is_refused_for_paired_end():
    with pytest.raises(miso_amd.InternalError):
        miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, collapsed=True)
