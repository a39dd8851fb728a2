"""GPU parity: the HIP sampler (through the C ABI) against the CPU oracle's counter mode.

Bar: bit-exact on everything integer (per-iteration per-isoform assignment counts, their FNV
hash, accepted/rejected, final assignment, classes) AND on the doubles (psi samples, log
scores), because host and device share include/miso_detmath.h + miso_philox.h.
"""
import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
import _problems
from _problems import simulate_pe, simulate_se

pytestmark = pytest.mark.gpu

SEED = 42


def _compare(gpu, cpu, C):
    assert (gpu.class_templates == cpu.class_templates).all()
    assert (gpu.class_counts == cpu.class_counts).all()
    assert (gpu.counts_trace == cpu.trace["counts_trace"]).all(), "assignment counts differ"
    assert (gpu.counts_hash == cpu.trace["counts_hash"]).all()
    assert gpu.rundata.noAccepted == cpu.accepted and gpu.rundata.noRejected == cpu.rejected
    assert (gpu.samples == cpu.samples).all(), np.abs(gpu.samples - cpu.samples).max()
    assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True)
    assert (gpu.assignment == cpu.assignment).all()
    assert gpu.rundata.noSamples == cpu.rundata[8]


@pytest.mark.parametrize("K,N,chains,iters,burn,lag", [
    (2, 1000, 1, 600, 100, 1), (2, 333, 3, 500, 50, 7), (3, 500, 2, 400, 100, 5),
    (5, 700, 2, 300, 50, 3), (10, 1000, 1, 200, 40, 2), (20, 600, 2, 100, 10, 3)])
def test_single_end_bit_exact(orc, K, N, chains, iters, burn, lag):
    exons, isoforms, g, pos, cig = simulate_se(orc, K, N, seed=100 + K)
    G = miso_amd.Gene(exons, isoforms)
    b = miso_amd.Batch(36, iters=iters, burn=burn, lag=lag, chains=chains, counts_trace=True)
    b.add_event(G, pos, cig)
    b.run(seed=SEED, first_event_id=5)
    gpu = b.result(0, trace=True)
    cpu = orc.miso(g, pos, cig, 36, iters=iters, burn=burn, lag=lag, chains=chains,
                   mode=OrcLib.COUNTER, seed=SEED, event_id=5, trace=True)
    assert cpu.rc == 0
    _compare(gpu, cpu, chains)


@pytest.mark.parametrize("K", list(range(3, 22)))
def test_single_end_kernel_of_every_isoform_count(orc, K, monkeypatch):
    """sampler_flat<KC, KS, UNI>: up to twenty isoforms the launch's largest isoform count is a compile-time constant of the
    kernel (slice layout, loop bounds -- kernels_flat.inl), and when every event of the launch has that count the kernel
    knows it (UNI: no per-chain `k < K` masks); MISO_FLAT_NO_UNI=1: the kernel for mixed counts; beyond twenty, and under
    MISO_FLAT_NO_KS=1, the run-time layout.  All bit-exact against the checker, three chains of different events in one
    wavefront."""
    iters, burn, lag, chains = 60, 10, 2, 2
    cases = []
    for e in range(3):
        exons, isoforms, g, pos, cig = simulate_se(orc, K, 250 + 40 * e, seed=700 + 10 * K + e)
        cases.append((miso_amd.Gene(exons, isoforms), g, pos, cig))
    kc = 4 if K <= 4 else 8 if K <= 8 else 12 if K <= 12 else 16 if K <= 16 else 32
    for env, name in ((None, "sampler_flat<%d, %d, true>" % (kc, K) if K <= 20 else "sampler_flat<%d, 0>" % kc),
                      ("MISO_FLAT_NO_UNI", "sampler_flat<%d, %d>" % (kc, K if K <= 20 else 0)),
                      ("MISO_FLAT_NO_KS", "sampler_flat<%d, 0>" % kc)):
        if env:
            monkeypatch.setenv(env, "1")
        b = miso_amd.Batch(36, iters=iters, burn=burn, lag=lag, chains=chains, counts_trace=True)
        for G, g, pos, cig in cases:
            b.add_event(G, pos, cig)
        b.run(seed=SEED, first_event_id=40)
        assert name in b.last_kernels(), b.last_kernels()
        for e, (G, g, pos, cig) in enumerate(cases):
            cpu = orc.miso(g, pos, cig, 36, iters=iters, burn=burn, lag=lag, chains=chains,
                           mode=OrcLib.COUNTER, seed=SEED, event_id=40 + e, trace=True)
            assert cpu.rc == 0
            _compare(b.result(e, trace=True), cpu, chains)


@pytest.mark.parametrize("K,N,chains,iters,burn,lag", [
    (2, 500, 1, 400, 100, 1), (2, 300, 3, 300, 50, 4), (3, 400, 2, 300, 50, 5), (5, 300, 1, 200, 20, 2)])
def test_paired_end_bit_exact(orc, K, N, chains, iters, burn, lag):
    exons, isoforms, g, pos, cig = simulate_pe(orc, K, N, seed=200 + K)
    G = miso_amd.Gene(exons, isoforms)
    b = miso_amd.Batch(36, iters=iters, burn=burn, lag=lag, chains=chains, paired=True, mean=250.0,
                       var=900.0, counts_trace=True)
    b.add_event(G, pos, cig)
    b.run(seed=SEED, first_event_id=9)
    gpu = b.result(0, trace=True)
    cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, iters=iters, burn=burn, lag=lag,
                          chains=chains, mode=OrcLib.COUNTER, seed=SEED, event_id=9, trace=True)
    assert cpu.rc == 0
    _compare(gpu, cpu, chains)


def test_many_events_one_launch(orc):
    """A mixed batch: event i of the batch == the oracle run with event_id = first + i."""
    specs = [(2, 400), (3, 250), (2, 64), (4, 300), (2, 1), (2, 0), (6, 500), (2, 257)]
    b = miso_amd.Batch(36, iters=300, burn=60, lag=3, chains=2)
    keep = []
    for i, (K, N) in enumerate(specs):
        exons, isoforms, g, pos, cig = simulate_se(orc, K, max(N, 1), seed=300 + i)
        pos, cig = pos[:N], cig[:N]
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        keep.append((g, pos, cig))
    b.run(seed=7, first_event_id=1000)
    for i, (g, pos, cig) in enumerate(keep):
        gpu = b.result(i)
        cpu = orc.miso(g, pos, cig, 36, iters=300, burn=60, lag=3, chains=2, mode=OrcLib.COUNTER,
                       seed=7, event_id=1000 + i, trace=True)
        assert (gpu.counts_hash == cpu.trace["counts_hash"]).all(), i
        assert (gpu.samples == cpu.samples).all(), i
        assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True), i
        assert (gpu.assignment == cpu.assignment).all(), i


@pytest.mark.parametrize("counts", [(3, 4, 3), (5, 8, 6, 7, 5), (9, 12, 10), (13, 16), (17, 20, 18)])
def test_single_end_class_of_mixed_isoform_counts(orc, counts):
    """Events of different isoform counts of one class in one launch share wavefronts: the kernel of the largest count,
    NOT the one that takes every chain's count for that constant (kernels_flat.inl UNI)."""
    iters, burn, lag, chains = 50, 10, 2, 2
    b = miso_amd.Batch(36, iters=iters, burn=burn, lag=lag, chains=chains, counts_trace=True)
    keep = []
    for e, K in enumerate(counts):
        exons, isoforms, g, pos, cig = simulate_se(orc, K, 200 + 30 * e, seed=9100 + 10 * K + e)
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        keep.append((g, pos, cig))
    b.run(seed=SEED, first_event_id=70)
    kc = 4 if max(counts) <= 4 else 8 if max(counts) <= 8 else 12 if max(counts) <= 12 else 16 if max(counts) <= 16 else 32
    assert b.last_kernels() == "sampler_flat<%d, %d>" % (kc, max(counts)), b.last_kernels()
    for e, (g, pos, cig) in enumerate(keep):
        cpu = orc.miso(g, pos, cig, 36, iters=iters, burn=burn, lag=lag, chains=chains,
                       mode=OrcLib.COUNTER, seed=SEED, event_id=70 + e, trace=True)
        assert cpu.rc == 0
        _compare(b.result(e, trace=True), cpu, chains)


@pytest.mark.parametrize("paired", [False, True])
def test_general_kernel_variants_agree(orc, paired):
    """sampler_grp with 2..32 lanes per chain and sampler_wave (64) are the same function."""
    import os
    specs = [(3, 300), (5, 257), (8, 400), (3, 5), (12, 350), (4, 0), (6, 64)]
    kw = dict(iters=150, burn=30, lag=2, chains=3)
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0,
                       counts_trace=True, **kw)
    keep = []
    for i, (K, N) in enumerate(specs):
        if paired:
            exons, isoforms, g, pos, cig = simulate_pe(orc, K, max(N, 1), seed=400 + i)
            pos, cig = pos[:2 * N], cig[:2 * N]
        else:
            exons, isoforms, g, pos, cig = simulate_se(orc, K, max(N, 1), seed=400 + i)
            pos, cig = pos[:N], cig[:N]
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        keep.append((g, pos, cig))
    old = os.environ.pop("MISO_GENERAL_LANES", None)
    try:
        ref = None
        for lanes in ("64", "2", "4", "8", "16", "32", None, "8-nocls"):
            os.environ.pop("MISO_NO_CLASS_PATH", None)
            if lanes is None:
                os.environ.pop("MISO_GENERAL_LANES", None)
            elif lanes.endswith("-nocls"):
                os.environ["MISO_GENERAL_LANES"] = lanes.split("-")[0]
                os.environ["MISO_NO_CLASS_PATH"] = "1"
            else:
                os.environ["MISO_GENERAL_LANES"] = lanes
            b.run(seed=21, first_event_id=3)
            got = [b.result(i, trace=True) for i in range(len(specs))]
            if ref is None:
                ref = got
                continue
            for x, y in zip(ref, got):
                assert np.array_equal(x.counts_trace, y.counts_trace), lanes
                assert np.array_equal(x.samples, y.samples), lanes
                assert np.array_equal(x.loglik, y.loglik, equal_nan=True), lanes
                assert np.array_equal(x.assignment, y.assignment), lanes
                assert x.rundata.noAccepted == y.rundata.noAccepted, lanes
    finally:
        os.environ.pop("MISO_GENERAL_LANES", None)
        os.environ.pop("MISO_NO_CLASS_PATH", None)
        if old is not None:
            os.environ["MISO_GENERAL_LANES"] = old
    for i, (g, pos, cig) in enumerate(keep):   # and the wave kernel's answer is the oracle's
        if paired:
            cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=21,
                                  event_id=3 + i, trace=True, **kw)
        else:
            cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=21, event_id=3 + i, trace=True, **kw)
        assert np.array_equal(ref[i].counts_trace, cpu.trace["counts_trace"]), i
        assert np.array_equal(ref[i].samples, cpu.samples), i


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("K", [2, 4])
def test_options_hyper_start_overhang(orc, K, paired):
    """Non-default options of the reference signature: Dirichlet hyper-parameters != 1
    (miso.c:165-182), START_UNIFORM (miso.c:372-387), overhang > 1 (miso.c:781)."""
    hyper = list(np.linspace(0.7, 2.5, K))
    kw = dict(iters=300, burn=60, lag=3, chains=2, overhang=3, start=1)
    if paired:
        exons, isoforms, g, pos, cig = simulate_pe(orc, K, 300, seed=500 + K)
        b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, counts_trace=True, **kw)
        cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, hyper=hyper, mode=OrcLib.COUNTER, seed=8,
                              event_id=2, trace=True, **kw)
    else:
        exons, isoforms, g, pos, cig = simulate_se(orc, K, 400, seed=500 + K)
        b = miso_amd.Batch(36, counts_trace=True, **kw)
        cpu = orc.miso(g, pos, cig, 36, hyper=hyper, mode=OrcLib.COUNTER, seed=8, event_id=2,
                       trace=True, **kw)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig, hyper=hyper)
    b.run(seed=8, first_event_id=2)
    _compare(b.result(0, trace=True), cpu, 2)


def test_many_read_classes_fall_back_to_masks(orc):
    """More than 32 distinct compatibility masks among the drawing reads: the class-threshold path
    does not apply, the direct mask path must give the oracle's answer."""
    K = 7
    exons = [(1 + 300 * i, 200 + 300 * i) for i in range(K + 1)]
    rng = np.random.default_rng(5)
    isoforms = [sorted(set([0, K]) | set(int(e) for e in rng.choice(np.arange(1, K), size=rng.integers(2, K - 1), replace=False)))
                for _ in range(K)]
    g = orc.gene([c for e in exons for c in e], isoforms)
    orc.rng_seed(77)
    rc, iso, pos, cig = orc.simulate_reads(g, np.ones(K) / K, 1500, 36)
    assert rc == 0
    probe = orc.miso(g, pos, cig, 36, iters=5, burn=1, lag=1, chains=1)
    kw = dict(iters=200, burn=40, lag=2, chains=2)
    b = miso_amd.Batch(36, counts_trace=True, **kw)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=13, first_event_id=1)
    cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=13, event_id=1, trace=True, **kw)
    _compare(b.result(0, trace=True), cpu, 2)
    assert len(probe.class_counts) >= 8


def test_large_event_and_thirty_two_isoforms(orc):
    """An event with 60 000 reads (draw list far beyond one trip per lane) and a 32-isoform gene."""
    exons, isoforms, g, pos, cig = simulate_se(orc, 2, 60000, seed=900)
    b = miso_amd.Batch(36, iters=60, burn=10, lag=1, chains=1, counts_trace=True)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    exons32, isoforms32, g32, pos32, cig32 = simulate_se(orc, 32, 500, seed=901, exlen=60, gap=50)
    b.add_event(miso_amd.Gene(exons32, isoforms32), pos32, cig32)
    b.run(seed=3, first_event_id=0)
    cpu = orc.miso(g, pos, cig, 36, iters=60, burn=10, lag=1, chains=1, mode=OrcLib.COUNTER, seed=3,
                   event_id=0, trace=True)
    _compare(b.result(0, trace=True), cpu, 1)
    cpu32 = orc.miso(g32, pos32, cig32, 36, iters=60, burn=10, lag=1, chains=1, mode=OrcLib.COUNTER,
                     seed=3, event_id=1, trace=True)
    _compare(b.result(1, trace=True), cpu32, 1)
    with pytest.raises(NotImplementedError, match="More than 256 isoforms"):
        e257, i257 = _problems.se_gene(257, exlen=60, gap=50)
        miso_amd.Batch(36).add_event(miso_amd.Gene(e257, i257), pos32[:5], cig32[:5])
    with pytest.raises(NotImplementedError, match="More than 256 isoforms"):
        miso_amd.Batch(36, device_match=True).add_event(miso_amd.Gene(e257, i257), pos32[:5], cig32[:5])


@pytest.mark.parametrize("K,paired,device_match", [(33, False, False), (33, False, True), (48, False, True), (64, False, True),
                                                   (64, False, False), (33, True, True), (48, True, False), (64, True, True)])
def test_more_than_thirty_two_isoforms_bit_exact(orc, K, paired, device_match):
    """Genes of 33 ... 64 isoforms (whole-gene mode on a real annotation has them; the reference has no limit, miso.c:696,
    gff.c:684; rounds 1 - 3 skipped them): a read's compatibility mask has two words, the chain runs on one wavefront with
    lane k = isoform k (sampler_wave).  Host and device matching, single- and paired-end, in a batch with smaller genes;
    every output against the oracle's counter mode."""
    kw = dict(iters=50, burn=10, lag=2, chains=2)
    evs = []
    for j, (k, n) in enumerate([(K, 400), (5, 300), (K, 90), (2, 200)]):
        if paired:
            exons, isoforms, g, pos, cig = simulate_pe(orc, k, n, seed=700 + j, exlen=420, gap=250)
        else:
            exons, isoforms, g, pos, cig = simulate_se(orc, k, n, seed=700 + j, exlen=60, gap=50)
        evs.append((exons, isoforms, g, pos, cig))
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0, counts_trace=True,
                       device_match=device_match, **kw)
    for exons, isoforms, g, pos, cig in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=SEED, first_event_id=40)
    assert "sampler_wave" in b.last_kernels(), b.last_kernels()
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        if paired:
            cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=SEED, event_id=40 + i, trace=True, **kw)
        else:
            cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=SEED, event_id=40 + i, trace=True, **kw)
        assert cpu.rc == 0
        _compare(b.result(i, trace=True), cpu, 2)
        if device_match and i == 0:   # what the match kernel wrote for the wide gene = the oracle's match matrix
            m, fl = b.device_match_of(0)
            assert np.array_equal(m != 0, cpu.match != 0)


@pytest.mark.parametrize("K,paired,device_match", [(65, False, False), (65, False, True), (100, False, True), (200, False, False),
                                                   (256, False, True), (65, True, True), (100, True, False), (200, True, True)])
def test_more_than_sixty_four_isoforms_bit_exact(orc, K, paired, device_match):
    """Genes of 65 ... 256 isoforms (the reference has no limit, miso.c:696, gff.c:684; rounds 1 - 4 stopped at 64): a read's
    compatibility mask has (K + 31) / 32 words -- host packing, the match kernel, the upload's planes -- and the chain's vectors
    live in LDS (sampler_big, kernels_big.hip).  Host and device matching, single- and paired-end, in a batch with smaller
    genes (one of them of 40 isoforms: sampler_wave beside it); every output against the oracle's counter mode, the read
    classes against the oracle's (= the real reference's, tests/test_host_logic.py)."""
    kw = dict(iters=40, burn=10, lag=2, chains=2)
    evs = []
    for j, (k, n) in enumerate([(K, 500), (5, 300), (K, 70), (40, 200), (K, 0)]):
        if paired:
            exons, isoforms, g, pos, cig = simulate_pe(orc, k, max(n, 1), seed=800 + j, exlen=420, gap=250)
        else:
            exons, isoforms, g, pos, cig = simulate_se(orc, k, max(n, 1), seed=800 + j, exlen=60, gap=50)
        evs.append((exons, isoforms, g, pos[:(2 * n if paired else n)], cig[:(2 * n if paired else n)]))
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0, counts_trace=True,
                       device_match=device_match, **kw)
    for exons, isoforms, g, pos, cig in evs:
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=SEED, first_event_id=90)
    assert "sampler_big" in b.last_kernels() and "sampler_wave" in b.last_kernels(), b.last_kernels()
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        if paired:
            cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=SEED, event_id=90 + i, trace=True, **kw)
        else:
            cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=SEED, event_id=90 + i, trace=True, **kw)
        assert cpu.rc == 0
        _compare(b.result(i, trace=True), cpu, 2)
        ct, cc = b.classes(i)
        assert np.array_equal(ct, cpu.class_templates) and np.array_equal(cc, cpu.class_counts)
        if i == 0 and not paired:
            # ADVICE r5 -- what bit parity alone would hide.  The reference evaluates its proposal density in the linear
            # domain (miso.c:97-122): beyond about 85 isoforms pow(2 pi sigma, -(K-1)/2) / prod(theta) overflows, the
            # Metropolis-Hastings ratio is inf - inf = NaN and `u < NaN` rejects: every chain accepts its first proposal
            # (iteration 0 leaves the proposal terms out, miso.c:866) and nothing after it.  Equal to the reference bit
            # for bit, and not a posterior: README.md / INTEGRATION.md say so, the library warns at upload.
            acc = b.result(0).rundata.noAccepted
            assert cpu.accepted == acc
            assert (acc > kw["chains"] * 10) if K <= 65 else (acc <= kw["chains"]), (K, acc)
        if device_match and i == 0:   # what the match kernel wrote for the wide gene = the oracle's match matrix
            m, fl = b.device_match_of(0)
            assert np.array_equal(m != 0, cpu.match != 0)


def test_two_isoform_reads_whose_high_half_sits_on_the_threshold(orc):
    """Single-end events with two isoforms draw a read's uniform in two half-words (include/miso_philox.h, lazy low bits):
    the device decides from the high halves, eight reads per Philox block, and draws a block's low halves only when a
    high half EQUALS the threshold's -- one read in 65 536.  Enough read-draws here (3.4e7) for some five hundred such
    reads, half of which change a count, on every lane layout (1 .. 64 lanes per chain and a workgroup per chain), with
    read counts that leave partial blocks; then the same batch with every lane forced through the rescan that serves a
    lane with SEVERAL such blocks in one step (MISO_K2_SETTLE_ALL=1): all bit for bit the checker's, which assembles
    both halves for every read."""
    import os
    kw = dict(iters=1200, burn=200, lag=5, chains=4)
    sizes = [3001, 1999, 1203, 777, 250, 61, 9, 8, 7, 1]
    b = miso_amd.Batch(36, counts_trace=True, **kw)
    keep = []
    for i, n in enumerate(sizes):
        exons, isoforms, g, pos, cig = simulate_se(orc, 2, n, seed=700 + i)
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        keep.append((g, pos, cig))
    cpu = [orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=31, event_id=40 + i, trace=True, **kw)
           for i, (g, pos, cig) in enumerate(keep)]
    saved = {k: os.environ.pop(k, None) for k in ("MISO_LANES_PER_CHAIN", "MISO_K2_SETTLE_ALL", "MISO_K2_TARGET")}
    try:
        # (the planner's widths; a bound on a wavefront's step so small that the largest events take a whole workgroup;
        # single-width launches)
        for env in (dict(), dict(MISO_K2_SETTLE_ALL="1"), dict(MISO_K2_TARGET="900"), dict(MISO_K2_TARGET="900", MISO_K2_SETTLE_ALL="1"),
                    dict(MISO_LANES_PER_CHAIN="1"), dict(MISO_LANES_PER_CHAIN="3", MISO_K2_SETTLE_ALL="1"),
                    dict(MISO_LANES_PER_CHAIN="8"), dict(MISO_LANES_PER_CHAIN="64")):
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(env)
            b.run(seed=31, first_event_id=40)
            for i in range(len(sizes)):
                got = b.result(i, trace=True)
                assert np.array_equal(got.counts_trace, cpu[i].trace["counts_trace"]), (env, i)
                assert np.array_equal(got.samples, cpu[i].samples), (env, i)
                assert np.array_equal(got.assignment, cpu[i].assignment), (env, i)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_two_isoform_chain_of_150000_reads_on_one_lane(orc):
    """A lane of the two-isoform read loop counts in two 16-bit halves of one register (kernels_k2.inl) and empties them
    every 16 000 generator blocks: 150 001 reads forced onto ONE lane (MISO_LANES_PER_CHAIN=1) are 18 750 blocks."""
    import os
    kw = dict(iters=12, burn=2, lag=1, chains=2)
    exons, isoforms, g, pos, cig = simulate_se(orc, 2, 150001, seed=811)
    b = miso_amd.Batch(36, counts_trace=True, **kw)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=3, event_id=9, trace=True, **kw)
    old = os.environ.pop("MISO_LANES_PER_CHAIN", None)
    try:
        for lanes in ("1", None):
            if lanes:
                os.environ["MISO_LANES_PER_CHAIN"] = lanes
            else:
                os.environ.pop("MISO_LANES_PER_CHAIN", None)
            b.run(seed=3, first_event_id=9)
            got = b.result(0, trace=True)
            assert np.array_equal(got.counts_trace, cpu.trace["counts_trace"]), lanes
            assert np.array_equal(got.samples, cpu.samples) and np.array_equal(got.assignment, cpu.assignment), lanes
    finally:
        os.environ.pop("MISO_LANES_PER_CHAIN", None)
        if old is not None:
            os.environ["MISO_LANES_PER_CHAIN"] = old


@pytest.mark.parametrize("event,lanes,iters", [(1444, "4", 200), (1007, "3", 1400)])
def test_a_read_on_the_threshold_in_a_full_block_and_one_in_the_partial_block_of_the_same_lane(orc, event, lanes, iters):
    """Regression (found by tests/test_gpu_scale.py at full size, one chain-step in 10^7): the chain's partial generator
    block is also the non-owned tail of one of its lane's regular trips, so a high half on the threshold in it flags
    that trip too -- two flagged trips, the lane rescans its full blocks -- and the partial block must then be settled
    once, not by the rescan as well.  Events of the benchmark's generator and the lane widths at which it happened."""
    import os
    from miso_amd import workload
    exons, isoforms, pos, cig = workload.event_reads(event, 2, 1000, 36, False, 0.0, 0.0)
    kw = dict(iters=iters, burn=0, lag=10, chains=1)
    b = miso_amd.Batch(36, counts_trace=True, **kw)
    b.set_event_id(b.add_event(miso_amd.Gene(exons, isoforms), pos, cig), event)
    g = orc.gene(_problems.flat(exons), isoforms)
    cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=42, event_id=event, trace=True, **kw)
    old = os.environ.pop("MISO_LANES_PER_CHAIN", None)
    try:
        os.environ["MISO_LANES_PER_CHAIN"] = lanes
        b.run(seed=42, first_event_id=0)
    finally:
        os.environ.pop("MISO_LANES_PER_CHAIN", None)
        if old is not None:
            os.environ["MISO_LANES_PER_CHAIN"] = old
    got = b.result(0, trace=True)
    assert np.array_equal(got.counts_trace, cpu.trace["counts_trace"])
    assert np.array_equal(got.samples, cpu.samples)
