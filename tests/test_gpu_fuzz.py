"""GPU: randomised mixed batches against the oracle's counter mode, bit for bit -- random isoform
counts (2-20, incl. the 17-20 range of the K <= 32 kernels), read counts (0-2500), chains, lags,
single- and paired-end, many events per launch so that several kernels (two-isoform + one per
isoform-count class) run concurrently on their own streams."""
import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
from _problems import expr_for, flat

pytestmark = pytest.mark.gpu


def random_gene(rng, K, exlen, gap):
    """K+1 exons; isoform 0 keeps all, the others drop a random non-empty subset of the inner exons
    (distinct isoforms) -- more compatibility classes than the skip-one-exon genes of _problems."""
    exons, s = [], 1
    for _ in range(K + 1):
        ln = int(rng.integers(exlen, 2 * exlen))
        exons.append((s, s + ln - 1))
        s += ln + gap
    seen, isoforms = {tuple(range(K + 1))}, [list(range(K + 1))]
    while len(isoforms) < K:
        drop = set(int(x) for x in rng.choice(np.arange(1, K), size=int(rng.integers(1, max(2, K // 2))), replace=False))
        iso = tuple(e for e in range(K + 1) if e not in drop)
        if iso not in seen:
            seen.add(iso)
            isoforms.append(list(iso))
    return exons, isoforms


@pytest.mark.parametrize("paired,seed", [(False, 1), (False, 2), (True, 3), (True, 4)])
def test_random_mixed_batches_bit_exact(orc, paired, seed):
    rng = np.random.default_rng(seed)
    chains, iters = int(rng.integers(1, 4)), int(rng.integers(60, 140))
    burn, lag = int(rng.integers(0, 30)), int(rng.integers(1, 6))
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains)
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0,
                       counts_trace=True, device_match=bool(seed % 2), **kw)
    cases = []
    ks = [2, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 18, 20] + [int(k) for k in rng.integers(2, 21, size=6)]
    for j, K in enumerate(ks):
        exons, isoforms = random_gene(rng, K, 400 if paired else 90, 300 if paired else 80)
        g = orc.gene(flat(exons), isoforms)
        n = int(rng.choice([0, 1, 3, 40, 400, 1200, 2500])) if j % 5 == 0 else int(rng.integers(50, 900))
        orc.rng_seed(1000 * seed + j)
        if paired:
            rc, iso, pos, cig = orc.simulate_paired_reads(g, expr_for(K), n, 36, 250.0, 900.0)
        else:
            rc, iso, pos, cig = orc.simulate_reads(g, expr_for(K), n, 36)
        assert rc == 0
        idx = b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        cases.append((idx, g, pos, cig, K))
    b.run(seed=99 + seed, first_event_id=50)
    for idx, g, pos, cig, K in cases:
        if paired:
            cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=99 + seed,
                                  event_id=50 + idx, trace=True, **kw)
        else:
            cpu = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=99 + seed, event_id=50 + idx,
                           trace=True, **kw)
        assert cpu.rc == 0
        gpu = b.result(idx, trace=True)
        assert (gpu.counts_hash == cpu.trace["counts_hash"]).all(), (K, len(pos))
        assert (gpu.counts_trace == cpu.trace["counts_trace"]).all(), (K, len(pos))
        assert np.array_equal(gpu.samples, cpu.samples, equal_nan=True), (K, len(pos))
        assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True)
        assert (gpu.assignment == cpu.assignment).all()
        assert gpu.rundata.noAccepted == cpu.accepted
    assert "," in b.last_kernels()          # several kernels in this launch


@pytest.mark.parametrize("level,seed", [(1, 11), (2, 12), (2, 13)])
def test_random_mixed_batches_collapsed_bit_exact(orc, level, seed):
    """The same random single-end batches with the collapsed Gibbs step against the checker's collapsed
    mode: level 1 collapses the two-isoform events only (the others run their per-read kernels = counter mode), level 2
    every event (sampler_lane_k: random genes have many more compatibility classes than skip-one-exon genes); random
    Dirichlet hyper-parameters on some events."""
    rng = np.random.default_rng(seed)
    chains, iters = int(rng.integers(1, 4)), int(rng.integers(60, 140))
    burn, lag = int(rng.integers(0, 30)), int(rng.integers(1, 6))
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains)
    b = miso_amd.Batch(36, counts_trace=True, device_match=bool(seed % 2), collapsed=level, **kw)
    cases = []
    ks = [2, 2, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20] + [int(k) for k in rng.integers(2, 12, size=6)]
    for j, K in enumerate(ks):
        exons, isoforms = random_gene(rng, K, 90, 80)
        g = orc.gene(flat(exons), isoforms)
        n = int(rng.choice([0, 1, 3, 40, 400, 1200, 6000])) if j % 4 == 0 else int(rng.integers(50, 900))
        orc.rng_seed(1000 * seed + j)
        rc, iso, pos, cig = orc.simulate_reads(g, expr_for(K), n, 36)
        assert rc == 0
        hyper = None if j % 3 else rng.uniform(0.5, 3.0, size=K)
        idx = b.add_event(miso_amd.Gene(exons, isoforms), pos, cig, hyper=hyper)
        cases.append((idx, g, pos, cig, K, hyper))
    b.run(seed=199 + seed, first_event_id=70)
    for idx, g, pos, cig, K, hyper in cases:
        mode = OrcLib.COLLAPSED if (K == 2 or level == 2) else OrcLib.COUNTER
        cpu = orc.miso(g, pos, cig, 36, hyper=hyper, mode=mode, seed=199 + seed, event_id=70 + idx, trace=True, **kw)
        assert cpu.rc == 0
        gpu = b.result(idx, trace=True)
        where = (level, K, len(pos))
        assert (gpu.counts_trace == cpu.trace["counts_trace"]).all(), where
        assert (gpu.counts_hash == cpu.trace["counts_hash"]).all(), where
        assert np.array_equal(gpu.samples, cpu.samples.reshape(gpu.samples.shape), equal_nan=True), where
        assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True), where
        assert (gpu.assignment == cpu.assignment).all(), where
        assert gpu.rundata.noAccepted == cpu.accepted, where
    names = b.last_kernels().split(",")
    assert ("sampler_lane" in names or "sampler_lane_ilp" in names) and (("sampler_lane_k" in names) == (level == 2)), names
