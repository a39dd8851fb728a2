"""GPU: algorithm = MARGINAL (miso.c:272-283, 800-808, 936-946) -- sampler_marginal (csrc/kernels_marginal.hip).

Bit for bit against the CPU checker's counter mode, whose stream mode replays the real reference's MARGINAL runs
(tests/test_oracle_golden.py::test_marginal_algorithm_golden, tests/golden/*_marginal*.npz); statistically against those
runs of the reference themselves."""
import numpy as np
import pytest

import _golden
import miso_amd
from miso_amd import capi
from _libs import OrcLib
from _problems import expr_for, flat, se_gene

pytestmark = pytest.mark.gpu
MARGINAL = capi.MISO_ALGO_MARGINAL


def _equal(gpu, cpu):
    assert cpu.rc == 0
    assert np.array_equal(gpu.samples, cpu.samples)
    assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True)
    assert np.array_equal(gpu.assignment, cpu.assignment)
    assert (gpu.rundata.noAccepted, gpu.rundata.noRejected) == (cpu.accepted, cpu.rejected)


@pytest.mark.parametrize("name", _golden.names("se_marginal"))
def test_golden_inputs_bit_exact(orc, name):
    g = _golden.load(name)
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    kw = dict(iters=g["iters"], burn=g["burn"], lag=g["lag"], chains=g["chains"], overhang=g["overhang"], algo=MARGINAL,
              stop=g["stop"], max_iters=g["max_iters"])
    b = miso_amd.Batch(g["read_len"], counts_trace=True, **kw)
    b.add_event(G, g["pos"], g["cigars"])
    b.run(seed=11, first_event_id=77)
    assert b.last_kernels().split(",")[0] == "sampler_marginal"
    gpu = b.result(0, trace=True)
    cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], mode=OrcLib.COUNTER, seed=11, event_id=77, trace=True, **kw)
    _equal(gpu, cpu)
    if g["stop"] == 0:
        assert np.array_equal(gpu.counts_hash, cpu.trace["counts_hash"])
        assert not gpu.counts_trace.any()
    else:
        assert b.rounds() > 1
    # read classes as the REAL reference returned them
    assert np.array_equal(gpu.class_templates, g["class_templates"])
    assert np.array_equal(gpu.class_counts, g["class_counts"])
    # the posterior against the reference's own MARGINAL run (another RNG): within 5 sigma of the Monte-Carlo error
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    a, r = gpu.samples[:filled], g["samples"][:filled]
    ess = max(filled / 20.0, 4.0)
    tol = 5 * np.sqrt((a.var(0) + r.var(0)) / ess) + 5e-3
    assert (np.abs(a.mean(0) - r.mean(0)) < tol).all(), (a.mean(0), r.mean(0), tol)


@pytest.mark.parametrize("device_match", [False, True])
def test_batch_of_mixed_events_bit_exact(orc, device_match):
    """Thirty events of 2 .. 40 isoforms (beyond 32: two mask words, 32 lanes per workgroup), 0 .. 800 reads, uniform
    and automatic start, host- and device-side matching: every event equals the checker's run of it alone."""
    rng = np.random.default_rng(8)
    for start in (capi.MISO_START_AUTO, capi.MISO_START_UNIFORM):
        kw = dict(iters=160, burn=40, lag=3, chains=3, algo=MARGINAL, start=start)
        b = miso_amd.Batch(36, device_match=device_match, **kw)
        cases = []
        for e in range(30):
            K = int(rng.choice([2, 2, 3, 4, 5, 7, 10, 16, 25, 33, 40]))
            exons, isoforms = se_gene(K)
            og = orc.gene(flat(exons), isoforms)
            orc.rng_seed(500 + e)
            n = int(rng.choice([0, 1, 5, 60, 300, 800]))
            rc, _, pos, cig = orc.simulate_reads(og, expr_for(K), max(n, 1), 36)
            assert rc == 0
            pos, cig = pos[:n], cig[:n]
            if n == 0:
                continue
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            cases.append((og, pos, cig))
        b.run(seed=21, first_event_id=300)
        assert b.last_kernels() == "sampler_marginal"
        for e, (og, pos, cig) in enumerate(cases):
            cpu = orc.miso(og, pos, cig, 36, mode=OrcLib.COUNTER, seed=21, event_id=300 + e, **kw)
            _equal(b.result(e), cpu)


def test_marginal_differs_from_reassign_and_needs_single_end(orc):
    g = _golden.load("se_k3_marginal")
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    out = []
    for algo in (capi.MISO_ALGO_REASSIGN, MARGINAL):
        b = miso_amd.Batch(g["read_len"], iters=600, burn=100, lag=2, chains=2, algo=algo)
        b.add_event(G, g["pos"], g["cigars"])
        b.run(seed=3)
        out.append(b.result(0).samples.mean(0))
    assert np.abs(out[0] - out[1]).max() > 1e-3      # another model (miso.c:272-283 has no length-normalised prior)


@pytest.mark.parametrize("device_match", [False, True])
def test_classes_algorithm_bit_exact(orc, device_match):
    """algorithm = CLASSES (miso.c:284-295, 788-803): the gene's possible read classes with the reads that fall into
    them -- the host's table (csrc/host.cpp attach_gene_classes) through sampler_marginal against the checker's run,
    whose matrix and score are pinned to the reference's own functions (tests/test_oracle_vs_ref.py; the reference's
    sampler itself reads its per-class counts uninitialised, miso.c:790).  Twenty events of 2 .. 40 isoforms, also under
    stop = CONVERGENT_MEAN; a caller-made problem has no gene structure to enumerate: NotImplementedError."""
    rng = np.random.default_rng(12)
    for stop, iters, burn in ((0, 160, 40), (1, 90, 30)):
        kw = dict(iters=iters, burn=burn, lag=3, chains=3, algo=capi.MISO_ALGO_CLASSES, stop=stop, max_iters=1500)
        b = miso_amd.Batch(36, device_match=device_match, **kw)
        cases = []
        for e in range(20):
            K = int(rng.choice([2, 2, 3, 4, 5, 7, 10, 16, 33, 40]))
            exons, isoforms = se_gene(K)
            og = orc.gene(flat(exons), isoforms)
            orc.rng_seed(600 + e)
            n = int(rng.choice([1, 5, 60, 300, 800]))
            rc, _, pos, cig = orc.simulate_reads(og, expr_for(K), n, 36)
            assert rc == 0
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            cases.append((og, pos, cig))
        b.run(seed=23, first_event_id=500)
        assert b.last_kernels().split(",")[0] == "sampler_marginal"
        for e, (og, pos, cig) in enumerate(cases):
            cpu = orc.miso(og, pos, cig, 36, mode=OrcLib.COUNTER, seed=23, event_id=500 + e, **kw)
            _equal(b.result(e), cpu)
    b = miso_amd.Batch(36, algo=capi.MISO_ALGO_CLASSES)
    with pytest.raises(NotImplementedError):
        b.add_problem(np.ones((3, 2)), isolen=[300, 200], noexons=[3, 2])


def test_the_module_call_takes_both_switches(orc):
    """pysplicing.MISO(gene, 0, positions, cigars, readLength, noIterations, noBurnIn, noLag, hyperp, overhang,
    no_chains, start, stop, algo) (pysplicing.c:41-131) with stop=CONVERGENT_MEAN and algo=MARGINAL: the tuple the
    reference's module returns, from the same run as the checker's."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miso_amd"))
    import pysplicing   # (the one module object tests/test_pysplicing_module.py and miso_sampler use)
    g = _golden.load("se_k3_marginal_convergent")
    gene = pysplicing.createGene(tuple(g["exon_list"]), tuple(tuple(i) for i in g["isoform_list"]))
    res = pysplicing.MISO(gene, 0, tuple(int(p) for p in g["pos"]), tuple(c.decode() for c in g["cigars"]), g["read_len"],
                          g["iters"], g["burn"], g["lag"], (1.0,) * 3, g["overhang"], g["chains"],
                          pysplicing.MISO_START_AUTO, pysplicing.MISO_STOP_CONVERGENT_MEAN, pysplicing.MISO_ALGO_MARGINAL,
                          seed=11)
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    # (the module's maxIterations is the reference's 100000, pysplicing.c:43)
    cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"], lag=g["lag"],
                   chains=g["chains"], overhang=g["overhang"], algo=1, stop=1, max_iters=100000, mode=OrcLib.COUNTER,
                   seed=11, event_id=0)
    assert cpu.rc == 0
    assert np.array_equal(np.transpose(np.array(res[0])), cpu.samples)
    assert np.array_equal(np.array(res[1]), cpu.loglik)
    assert list(res[4]) == list(cpu.assignment)
