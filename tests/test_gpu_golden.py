"""GPU: the HIP path on the committed golden INPUTS (incl. the reference's own Atp2b1 test data),
against the oracle's counter mode; plus the sampler entry points that mirror splicing_miso /
splicing_miso_paired one event at a time."""
import ctypes as C

import numpy as np
import pytest

import _golden
import miso_amd
from _libs import OrcLib
from _problems import flat
from miso_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", _golden.names("se") + _golden.names("pe"))
def test_golden_inputs_bit_exact(orc, name):
    g = _golden.load(name)
    paired = str(g["kind"]) == "pe"
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    kw = dict(iters=g["iters"], burn=g["burn"], lag=g["lag"], chains=g["chains"], overhang=g["overhang"])
    if paired:
        mean, var = float(g["mean"]), float(g["var"])
        b = miso_amd.Batch(g["read_len"], paired=True, mean=mean, var=var, counts_trace=True, **kw)
        cpu = orc.miso_paired(og, g["pos"], g["cigars"], g["read_len"], mean, var,
                              mode=OrcLib.COUNTER, seed=11, event_id=77, trace=True, **kw)
    else:
        b = miso_amd.Batch(g["read_len"], counts_trace=True, **kw)
        cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], mode=OrcLib.COUNTER, seed=11,
                       event_id=77, trace=True, **kw)
    b.add_event(G, g["pos"], g["cigars"])
    b.run(seed=11, first_event_id=77)
    gpu = b.result(0, trace=True)
    assert cpu.rc == 0
    # set-up quantities equal the REAL reference's (golden), not just the oracle's
    assert np.array_equal(gpu.class_templates, g["class_templates"])
    assert np.array_equal(gpu.class_counts, g["class_counts"])
    assert np.array_equal(gpu.counts_trace, cpu.trace["counts_trace"])
    assert np.array_equal(gpu.counts_hash, cpu.trace["counts_hash"])
    assert np.array_equal(gpu.samples, cpu.samples)
    assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True)
    assert np.array_equal(gpu.assignment, cpu.assignment)
    assert (gpu.rundata.noAccepted, gpu.rundata.noRejected) == (cpu.accepted, cpu.rejected)
    # and the posterior agrees with the reference run stored in the fixture (different RNG):
    # |delta mean psi| within 5 sigma of the two runs' Monte-Carlo error
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    a, r = gpu.samples[:filled], g["samples"][:filled]
    ess = max(filled / 20.0, 4.0)  # conservative effective sample size
    tol = 5 * np.sqrt((a.var(0) + r.var(0)) / ess) + 5e-3
    assert (np.abs(a.mean(0) - r.mean(0)) < tol).all(), (a.mean(0), r.mean(0), tol)


@pytest.mark.parametrize("K,paired,n,switches", [
    (5, True, 600, {}), (10, True, 700, {}), (10, False, 800, {}), (3, True, 500, {}), (5, False, 700, {}),
    (2, False, 600, {}), (2, True, 500, {}),
    # algorithm = MARGINAL (sampler_marginal) and stop = CONVERGENT_MEAN (the device re-runs the longer schedule from the
    # start where the reference continues its chains: the same law, tested as such)
    (3, False, 500, dict(algo=1)), (6, False, 700, dict(algo=1)),
    (2, False, 500, dict(stop=1, iters=200, burn=50, lag=2, chains=3, max_iters=3000)),
    (3, True, 400, dict(stop=1, iters=200, burn=50, lag=2, chains=3, max_iters=3000)),
    (4, False, 500, dict(algo=1, stop=1, iters=200, burn=50, lag=2, chains=3, max_iters=3000))])
def test_device_posterior_within_monte_carlo_error_of_the_real_reference(ref, K, paired, n, switches):
    """The GPU against the REAL reference C core (oracle/_ref, independent random streams), every kernel family, ALL
    isoforms, posterior mean and both Chen-Shao bounds -- as a TWO-sample test: 10 genes, each under 8 random streams on
    the GPU (8 copies of the gene with different ids in the Philox counter, one launch) and under 8 seeds of the
    reference; the two groups of runs are compared by the exact permutation tests of tests/_dpsi.py (pooled t^2, the
    largest |t|, signed shifts of the mean and of either bound, shrinkage, interval width; Bonferroni over the seven, row
    level 1e-3).  Round 3 held ONE device run against 4 x MCSE of 8 reference seeds: blind to any bias below ~1 MCSE.
    bench.py runs the same test at the benchmark's sizes for every matrix row.  Matches miso.c:845-900,
    miso_paired.c:451-498, credible_intervals.py:31-55."""
    import _dpsi
    from _problems import se_gene, expr_for
    kw = dict(dict(iters=3000, burn=1000, lag=2, chains=1), **switches)
    n_genes, n_streams = 10, 8
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0, device_match=True, **kw)
    probs = []
    for j in range(n_genes):
        exons, isoforms = se_gene(K, exlen=(500 if paired else 110) + 9 * j, gap=300 if paired else 100)
        g = ref.gene(flat(exons), isoforms)
        ref.rng_seed(8000 + j)
        if paired:
            rc, _, pos, cig = ref.simulate_paired_reads(g, expr_for(K), n, 36, 250.0, 900.0)
        else:
            rc, _, pos, cig = ref.simulate_reads(g, expr_for(K), n, 36)
        assert rc == 0
        probs.append((g, pos, cig, miso_amd.Gene(exons, isoforms)))
    for s in range(n_streams):
        for j, (_, pos, cig, G) in enumerate(probs):
            i = b.add_event(G, pos, cig)
            b.set_event_id(i, ((s + 1) << 12) + j)
    b.run(seed=77, first_event_id=0)
    b.summarize(0.95)
    gpu = {j: [b.summary(s * n_genes + j) for s in range(n_streams)] for j in range(n_genes)}

    def stats(samples):
        x = np.sort(samples, axis=0)
        m = len(x)
        return x.mean(0), x[int(round(0.025 * m)) - 1], x[int(round(0.975 * m)) - 1]
    cpu = {}
    for j, (g, pos, cig, _) in enumerate(probs):
        cpu[j] = []
        for s in range(8):
            ref.rng_seed(9000 + 17 * j + s)
            r = ref.miso_paired(g, pos, cig, 36, 250.0, 900.0, **kw) if paired else ref.miso(g, pos, cig, 36, **kw)
            assert r.rc == 0
            cpu[j].append(stats(r.samples))
    ev = list(range(n_genes))
    res = _dpsi.two_sample(_dpsi.stack_runs(gpu, ev, K), _dpsi.stack_runs(cpu, ev, K))
    assert res["pass"], (K, paired, res)


def test_per_event_entry_points(orc):
    """miso_run / miso_run_paired: the splicing_miso signatures + seed, a batch of one."""
    g = _golden.load("se_k3")
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    K, N, S = 3, len(g["pos"]), g["chains"] * (g["iters"] - g["burn"]) // g["lag"]
    samples, ll = np.zeros(K * S), np.zeros(S)
    ct, cc, ncls = np.zeros(K * N), np.zeros(N), C.c_int()
    ass, rd = np.zeros(N, np.int32), capi.RunData()
    pos = np.asarray(g["pos"], np.int32)
    rc = capi.lib().miso_run(G.handle, capi._p(pos), capi._cigs(g["cigars"]), N, g["read_len"],
                             g["overhang"], g["chains"], g["iters"], 100000, g["burn"], g["lag"],
                             capi._p(np.ones(K)), K, 0, 0, 0, C.c_uint64(5), capi._p(samples),
                             capi._p(ll), capi._p(ct), capi._p(cc), C.byref(ncls), capi._p(ass),
                             C.byref(rd))
    capi.check(rc)
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"],
                   lag=g["lag"], chains=g["chains"], overhang=g["overhang"], mode=OrcLib.COUNTER,
                   seed=5, event_id=0)
    assert np.array_equal(samples.reshape(S, K), cpu.samples) and np.array_equal(ll, cpu.loglik)
    assert np.array_equal(ass, cpu.assignment) and ncls.value == len(g["class_counts"])
    assert (rd.noIso, rd.noIters, rd.noBurnIn, rd.noLag, rd.noChains, rd.noSamples) == \
        (K, g["iters"], g["burn"], g["lag"], g["chains"], S)
    assert rd.noAccepted + rd.noRejected == g["chains"] * g["iters"]


def test_error_paths_on_device(orc):
    G = miso_amd.Gene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]])
    rc = capi.lib().miso_run(G.handle, capi._p(np.array([10], np.int32)), capi._cigs([b"36Q"]), 1, 36,
                             1, 1, 50, 100000, 10, 1, capi._p(np.ones(2)), 2, 0, 0, 0, C.c_uint64(1),
                             None, None, None, None, None, None, None)
    assert rc == capi.MISO_EINVAL and b"Unsupported CIGAR" in capi.lib().miso_last_error()
