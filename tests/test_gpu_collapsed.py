"""The collapsed Gibbs step (csrc/kernels_lane.hip, include/miso_binomial.h): single-end two-isoform events draw the
COUNT of their exchangeable reads on isoform 0 as one exact binomial per iteration instead of one uniform per read
(reference: miso.c:30-91 per read, then counts only, miso.c:243-307).  Bit for bit against the checker's collapsed
mode (oracle/miso_oracle.c ORC_MODE_COLLAPSED) -- samples, log scores, per-iteration counts, the final per-read
assignment -- on events from 0 to 60 000 reads in one launch; the binomial sampler itself and the agreement of the
collapsed chain with the per-read chain and the real reference are CPU tests (tests/test_collapsed.py)."""
import contextlib
import os

import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
from _problems import flat, se_gene, expr_for

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


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
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, seed=77, event_id=1200 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    # lanes per chain: 1 = sampler_lane; 2, 4, 8 = sampler_k2c (the binomial's rejection trials that many at a time,
    # the Metropolis-Hastings step's transcendentals one per lane): the same bits
    # one lane per chain has two forms (kernels_lane.hip): sampler_lane_ilp for batches of at most two wavefronts per SIMD
    # (this one), sampler_lane beyond (MISO_LANE_ILP forces either)
    for lanes, ilp, name in ((1, "1", "sampler_lane_ilp"), (1, "0", "sampler_lane"), (2, None, "sampler_k2c<2>"), (4, None, "sampler_k2c<4>"),
                             (8, None, "sampler_k2c<8>"), (None, None, "sampler_lane_ilp")):
        with _env(MISO_COLLAPSED_LANES=None if lanes is None else str(lanes), MISO_LANE_ILP=ilp):
            b = miso_amd.Batch(36, counts_trace=True, collapsed=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=77, first_event_id=1200)
        assert b.last_kernels() == name
        for i, r in enumerate(cpu):
            gpu = b.result(i, trace=True)
            where = (sizes[i], kw, name)
            assert np.array_equal(gpu.counts_trace, r.trace["counts_trace"]), where
            assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
            assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
            assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
            assert (gpu.assignment == r.assignment).all(), where
            assert gpu.rundata.noAccepted == r.accepted, where


@pytest.mark.parametrize("K,chains,iters,burn,lag", [(3, 1, 200, 40, 1), (5, 2, 150, 10, 3), (10, 1, 120, 20, 1), (18, 1, 60, 10, 1),
                                                     (4, 2, 0, 0, 1), (7, 1, 1, 0, 1)])
def test_collapsed_three_or_more_isoforms_bit_exact_against_the_checker(orc, K, chains, iters, burn, lag):
    """sampler_lane_k: per compatibility class a chain of binomials (include/miso_binomial.h), classes in the order of
    the event's class table; events of 0 .. 20 000 reads and of K and K - 1 isoforms in one launch."""
    sizes = [500, 20000, 12, 0, 3, 150, 4000, 45, 1000, 64, 2500, 31]
    evs = []
    for j, n in enumerate(sizes):
        Kj = K if j % 4 else max(3, K - 1)
        exons, isoforms = se_gene(Kj, exlen=200 + 13 * j)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(5000 + j)
        rc, _, pos, cig = orc.simulate_reads(g, expr_for(Kj), max(n, 1), 36)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:n], cig[:n], Kj))
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains)
    b = miso_amd.Batch(36, counts_trace=True, collapsed=2, **kw)
    for exons, isoforms, g, pos, cig, Kj in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=91, first_event_id=40)
    assert set(b.last_kernels().split(",")) == {"sampler_lane_k"}    # (one launch per isoform-count class since round 6)
    for i, (exons, isoforms, g, pos, cig, Kj) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, seed=91, event_id=40 + i, trace=True, **kw)
        assert r.rc == 0
        gpu = b.result(i, trace=True)
        where = (Kj, sizes[i], kw)
        assert np.array_equal(gpu.counts_trace, r.trace["counts_trace"]), where
        assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
        assert np.array_equal(gpu.samples, r.samples.reshape(gpu.samples.shape), equal_nan=True), where
        assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
        assert (gpu.assignment == r.assignment).all(), where
        assert gpu.rundata.noAccepted == r.accepted, where


def test_collapsed_mixed_isoform_counts_use_both_kernels(orc):
    evs = []
    for j, (K, n) in enumerate([(2, 300), (5, 400), (2, 50), (3, 800)]):
        exons, isoforms = se_gene(K, exlen=250 + 9 * j)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(6000 + j)
        rc, _, pos, cig = orc.simulate_reads(g, expr_for(K), n, 36)
        evs.append((exons, isoforms, g, pos, cig))
    kw = dict(iters=100, burn=20, lag=2, chains=2)
    b = miso_amd.Batch(36, collapsed=2, **kw)
    for exons, isoforms, g, pos, cig in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=5, first_event_id=7)
    assert set(b.last_kernels().split(",")) == {"sampler_lane_ilp", "sampler_lane_k"}   # (sampler_lane_k once per isoform-count class)
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, seed=5, event_id=7 + i, trace=True, **kw)
        gpu = b.result(i)
        assert np.array_equal(gpu.samples, r.samples.reshape(gpu.samples.shape), equal_nan=True), i
        assert (gpu.assignment == r.assignment).all(), i


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


def test_collapsed_results_do_not_depend_on_batching(orc):
    """Every draw is addressed by (seed, event id, chain, iteration): six events in one batch, in two batches of three
    with the matching first_event_id, and with explicit ids in another order give the same bits."""
    evs = _events(orc, [300, 50, 1200, 7, 640, 90])
    kw = dict(iters=150, burn=30, lag=2, chains=2)

    def batch(items, first):
        b = miso_amd.Batch(36, collapsed=True, **kw)
        for exons, isoforms, g, pos, cig in items:
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        b.run(seed=9, first_event_id=first)
        return b
    whole = batch(evs, 100)
    a, c = batch(evs[:3], 100), batch(evs[3:], 103)
    for i in range(6):
        part = a.result(i) if i < 3 else c.result(i - 3)
        assert np.array_equal(whole.result(i).samples, part.samples) and (whole.result(i).assignment == part.assignment).all()
    b = miso_amd.Batch(36, collapsed=True, **kw)
    order = [4, 1, 5, 0, 3, 2]
    for j, i in enumerate(order):
        exons, isoforms, g, pos, cig = evs[i]
        b.set_event_id(b.add_event(miso_amd.Gene(exons, isoforms), pos, cig), 100 + i)
    b.run(seed=9, first_event_id=0)
    for j, i in enumerate(order):
        assert np.array_equal(whole.result(i).samples, b.result(j).samples)


def test_collapsed_chains_dealt_over_all_simds(orc):
    """A batch of fewer wavefronts than the device has SIMDs gets fewer than 64 chains per wavefront (runtime.hip: the
    launch of sampler_lane_ilp; kernels_lane.hip lane_body): 9000 chains -> nine per wavefront.  Sixteen events repeated
    under their own ids: every copy equals the checker's run of the event, with the rule and without (MISO_LANE_SPREAD=0)."""
    sizes = [300, 20, 0, 3, 150, 45, 64, 65, 31, 1, 333, 250, 120, 7, 90, 200]
    evs = _events(orc, sizes)
    kw = dict(iters=60, burn=10, lag=2, chains=1)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COLLAPSED, seed=31, event_id=500 + i, **kw)
        assert r.rc == 0
        cpu.append(r)
    genes = [miso_amd.Gene(exons, isoforms) for exons, isoforms, g, pos, cig in evs]
    n = 9000
    for spread in (None, "0"):
        with _env(MISO_LANE_SPREAD=spread):
            b = miso_amd.Batch(36, collapsed=True, **kw)
            for j in range(n):
                i = j % 16
                b.set_event_id(b.add_event(genes[i], evs[i][3], evs[i][4]), 500 + i)
            b.run(seed=31, first_event_id=0)
        assert b.last_kernels() == "sampler_lane_ilp"
        for j in list(range(40)) + list(range(n - 40, n)) + list(range(1000, n, 997)):
            r, gpu = cpu[j % 16], b.result(j)
            assert np.array_equal(gpu.samples, r.samples, equal_nan=True), (j, spread)
            assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), (j, spread)
            assert (gpu.assignment == r.assignment).all(), (j, spread)
            assert gpu.rundata.noAccepted == r.accepted, (j, spread)


def test_collapsed_level_two_with_a_gene_of_forty_isoforms_and_no_ambiguous_read(orc):
    """a single-end gene of 33 - 64 isoforms none of whose reads is ambiguous kept `lane_gen` on (only events
    with drawing reads could switch it off) and sampler_lane_k -- 32-bit masks, an LDS slice beyond the CU's -- failed the
    whole batch.  Such genes take sampler_wave whatever the collapsed level."""
    K = 40
    exons = [(1 + 300 * k, 200 + 300 * k) for k in range(K)]
    isoforms = [[k] for k in range(K)]          # disjoint isoforms: every read has exactly one compatible isoform
    rng = np.random.default_rng(7)
    pos = np.array([1 + 300 * int(k) + int(o) for k, o in zip(rng.integers(0, K, 400), rng.integers(0, 160, 400))], np.int32)
    cig = [b"36M"] * len(pos)
    kw = dict(iters=120, burn=20, lag=2, chains=2)
    b = miso_amd.Batch(36, collapsed=2, **kw)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    ex2, iso2 = se_gene(3)
    g2 = orc.gene(flat(ex2), iso2)
    orc.rng_seed(77)
    rc, _, pos2, cig2 = orc.simulate_reads(g2, expr_for(3), 300, 36)
    b.add_event(miso_amd.Gene(ex2, iso2), pos2, cig2)
    b.run(seed=3, first_event_id=10)
    g = orc.gene(flat(exons), isoforms)
    for i, (gg, p_, c_) in enumerate([(g, pos, cig), (g2, pos2, cig2)]):
        r = orc.miso(gg, p_, c_, 36, mode=OrcLib.COLLAPSED if i else OrcLib.COUNTER, seed=3, event_id=10 + i, **kw)
        gpu = b.result(i)
        assert r.rc == 0
        # event 0, no drawing read: the collapsed and the per-read chain are the same chain.  Event 1 (round 6, ADVICE r5:
        # the route is decided per run): the three-isoform gene keeps its collapsed step beside the forty-isoform one
        assert np.array_equal(gpu.samples, r.samples.reshape(gpu.samples.shape), equal_nan=True), i
        assert (gpu.assignment == r.assignment).all(), i
    assert sorted(b.last_kernels().split(",")) == ["sampler_lane_k", "sampler_wave<false>"], b.last_kernels()


def test_collapsed_level_two_routes_every_run_on_its_own(orc):
    """one gene of 33 and more isoforms in a level-2 collapsed batch used to send EVERY general event back to the
    per-read kernels -- sampled in counter mode while the checker's COLLAPSED mode expects collapsed draws.  Now the classes
    that have their class tables take sampler_lane_k (= the checker's COLLAPSED mode, bit for bit) and only the wide gene
    is sampled per read (= the checker's COUNTER mode), ambiguous reads and all."""
    kw = dict(iters=150, burn=30, lag=2, chains=2)
    evs = []
    for j, (K, n) in enumerate([(3, 400), (40, 500), (5, 900), (2, 300), (10, 250), (36, 120)]):
        exons, isoforms = se_gene(K, exlen=220 + 7 * j)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(8100 + j)
        rc, _, pos, cig = orc.simulate_reads(g, expr_for(K), n, 36)
        assert rc == 0
        evs.append((K, exons, isoforms, g, pos, cig))
    b = miso_amd.Batch(36, collapsed=2, **kw)
    for K, exons, isoforms, g, pos, cig in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=17, first_event_id=300)
    kernels = b.last_kernels().split(",")
    assert "sampler_lane_k" in kernels and "sampler_wave<false>" in kernels and "sampler_lane_ilp" in kernels, kernels
    for i, (K, exons, isoforms, g, pos, cig) in enumerate(evs):
        mode = OrcLib.COUNTER if K > 32 else OrcLib.COLLAPSED
        r = orc.miso(g, pos, cig, 36, mode=mode, seed=17, event_id=300 + i, **kw)
        gpu = b.result(i)
        assert r.rc == 0
        assert np.array_equal(gpu.samples, r.samples.reshape(gpu.samples.shape), equal_nan=True), (i, K)
        assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), (i, K)
        assert (gpu.assignment == r.assignment).all(), (i, K)
        assert gpu.rundata.noAccepted == r.accepted, (i, K)
