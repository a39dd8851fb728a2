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
