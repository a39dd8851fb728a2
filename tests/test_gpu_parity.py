"""GPU parity: the HIP sampler (through the C ABI) against the CPU oracle's counter mode.

Bar: bit-exact on everything integer (per-iteration per-isoform assignment counts, their FNV
hash, accepted/rejected, final assignment, classes) AND on the doubles (psi samples, log
scores), because host and device share include/miso_detmath.h + miso_philox.h.
"""
import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
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
        for lanes in ("64", "2", "4", "8", "16", "32", None):
            if lanes is None:
                os.environ.pop("MISO_GENERAL_LANES", None)
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
