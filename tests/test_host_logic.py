"""CPU: the product's host side (gene model, CIGAR matching, read classes, parameter checks,
simulator) against the oracle and the golden vectors.  No GPU needed: this is the input builder
of the path (solve.c:8-306) plus the boundary's error behaviour (pysplicing.c / pyerror.c)."""
import numpy as np
import pytest

import _golden
import miso_amd
from miso_amd import capi, workload
from _problems import flat


@pytest.mark.parametrize("name", _golden.names("se"))
def test_match_matrix_and_classes_vs_golden(name):
    g = _golden.load(name)
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    m = G.match_iso(g["pos"], g["cigars"], g["read_len"], overhang=g["overhang"])
    assert np.array_equal(m, g["match"])
    b = miso_amd.Batch(g["read_len"], iters=g["iters"], burn=g["burn"], lag=g["lag"],
                       chains=g["chains"], overhang=g["overhang"])
    i = b.add_event(G, g["pos"], g["cigars"])
    ct, cc = b.classes(i)
    assert np.array_equal(ct, g["class_templates"]) and np.array_equal(cc, g["class_counts"])


@pytest.mark.parametrize("name", _golden.names("pe"))
def test_paired_match_and_binary_classes_vs_golden(name):
    g = _golden.load(name)
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    m, fl = G.match_iso_paired(g["pos"], g["cigars"], g["read_len"], float(g["mean"]), float(g["var"]))
    assert np.array_equal(m, g["match"]) and np.array_equal(fl, g["fraglen"])
    b = miso_amd.Batch(g["read_len"], iters=g["iters"], burn=g["burn"], lag=g["lag"],
                       chains=g["chains"], paired=True, mean=float(g["mean"]), var=float(g["var"]))
    i = b.add_event(G, g["pos"], g["cigars"])
    ct, cc = b.classes(i)
    assert np.array_equal(ct, g["class_templates"]) and np.array_equal(cc, g["class_counts"])


def test_cigar_edge_cases_vs_golden():
    g = _golden.load("cigar_edges")
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    for ov in (1, 4):
        assert np.array_equal(G.match_iso(g["pos"], g["cigars"], 36, overhang=ov), g["match_ov%d" % ov])


def test_isoform_lengths(orc):
    g = _golden.load("atp2b1")
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    assert G.noiso == 2 and (G.isolength() == orc.isolength(og)).all()


@pytest.mark.parametrize("bad,msg", [
    ([b"36Q"], "Unsupported CIGAR"), ([b"10M5S10M"], "Bad CIGAR string")])
def test_bad_cigar_raises_like_the_reference(bad, msg):
    G = miso_amd.Gene([(1, 100), (201, 300)], [[0, 1], [0]])
    with pytest.raises(miso_amd.InternalError, match=msg):
        G.match_iso([10], bad, 36)


def test_parameter_errors():
    # miso.c:674-717; what the GPU build does not implement raises NotImplementedError
    with pytest.raises(miso_amd.InternalError, match="Overhang length invalid"):
        miso_amd.Batch(36, overhang=18)
    with pytest.raises(miso_amd.InternalError, match="at least one"):
        miso_amd.Batch(36, chains=0)
    with pytest.raises(miso_amd.InternalError, match="one chain only"):
        miso_amd.Batch(36, chains=1, stop=capi.MISO_STOP_CONVERGENT_MEAN)
    miso_amd.Batch(36, chains=2, stop=capi.MISO_STOP_CONVERGENT_MEAN)
    with pytest.raises(miso_amd.InternalError, match="`stop` is invalid"):
        miso_amd.Batch(36, chains=2, stop=2)
    miso_amd.Batch(36, algo=capi.MISO_ALGO_CLASSES)
    with pytest.raises(NotImplementedError, match="Overhang is not implemented in assignment matrix"):
        miso_amd.Batch(36, algo=capi.MISO_ALGO_CLASSES, overhang=2)      # assignment.c:103-106
    miso_amd.Batch(36, algo=capi.MISO_ALGO_MARGINAL)
    with pytest.raises(miso_amd.InternalError, match="`algorithm` is invalid"):
        miso_amd.Batch(36, algo=3)
    with pytest.raises(miso_amd.InternalError, match="start_psi"):
        miso_amd.Batch(36, start=capi.MISO_START_GIVEN)
    G = miso_amd.Gene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]])
    b = miso_amd.Batch(36)
    with pytest.raises(miso_amd.InternalError, match="hyperparameter"):
        b.add_event(G, [10], [b"36M"], hyper=[1.0, 1.0, 1.0])
    with pytest.raises(miso_amd.InternalError):
        miso_amd.Gene([(1, 100)], [[0, 3]])


def test_product_simulator_is_deterministic_and_valid(orc):
    """miso_simulate_reads: reads of isoform k are compatible with isoform k (checked with the
    ORACLE's matcher), same seed -> same reads, batch ingestion == explicit ingestion."""
    for K, paired in ((2, False), (5, False), (2, True), (3, True)):
        kw = dict(min_len=400, max_len=800, gap=300) if paired else {}
        exons, isoforms, expr = workload.event_gene(17, K, **kw)
        G = miso_amd.Gene(exons, isoforms)
        mean, var = (250.0, 900.0) if paired else (0.0, 0.0)
        iso, pos, cig = capi.simulate_reads(G, expr, 200, 36, 99, mean, var)
        iso2, pos2, cig2 = capi.simulate_reads(G, expr, 200, 36, 99, mean, var)
        assert (pos == pos2).all() and cig == cig2
        og = orc.gene(flat(exons), isoforms)
        rc, m = orc.match_iso(og, pos, cig, 36)
        assert rc == 0 and (m[np.arange(len(pos)), iso] == 1).all()
        if paired:
            rc, mp, fl = orc.match_iso_paired(og, pos, cig, 36, mean, var)
            assert rc == 0 and (fl[np.arange(200), iso[::2]] >= 0).mean() > 0.99
        b1 = miso_amd.Batch(36, paired=paired, mean=mean, var=var)
        b2 = miso_amd.Batch(36, paired=paired, mean=mean, var=var)
        b1.add_simulated(G, expr, 200, 99)
        b2.add_event(G, pos, cig)
        assert all(np.array_equal(x, y) for x, y in zip(b1.classes(0), b2.classes(0)))


def test_workload_is_a_function_of_the_global_event_id():
    a = workload.build_batch(5, 3, n_reads=50, iters=10, burn=2)
    b = workload.build_batch(6, 1, n_reads=50, iters=10, burn=2)
    assert all(np.array_equal(x, y) for x, y in zip(a.classes(1), b.classes(0)))
    assert workload.shard_bounds(10, 4, 0) == (0, 3) and workload.shard_bounds(10, 4, 3) == (8, 10)
    cover = [i for r in range(7) for i in range(*workload.shard_bounds(40, 7, r))]
    assert cover == list(range(40))


def test_miso_file_number_formatting_equals_python():
    """miso_batch_write_miso_files prints "%.4f" / "%.2f" (miso_sampler.py:458-464) with a fast path;
    every digit must equal Python's % operator, ties and near-ties included."""
    import ctypes as C
    L = capi.lib()
    rng = np.random.default_rng(1)

    def mismatches(x, d):
        x = np.ascontiguousarray(x, dtype=np.float64)
        buf = C.create_string_buffer(344 * len(x))
        assert L.miso_selftest_format(x.ctypes.data_as(C.c_void_p), len(x), d, buf, 344) == 0
        raw = buf.raw
        fmt = "%%.%df" % d
        return [(v, raw[344 * i:344 * (i + 1)].split(b"\0")[0].decode(), fmt % v)
                for i, v in enumerate(x) if raw[344 * i:344 * (i + 1)].split(b"\0")[0].decode() != fmt % v]

    ties4 = (np.arange(0, 10000) + 0.5) / 10000.0
    psi = np.concatenate([rng.uniform(0, 1, 100000), np.arange(0, 10001) / 10000.0, ties4,
                          np.nextafter(ties4, 1), np.nextafter(ties4, 0),
                          [0.0, -0.0, 1.0, np.nan, np.inf, -np.inf, 1e-310, 0.99995, 5e-5, 4.9999999e-5]])
    assert mismatches(psi, 4) == []
    ties2 = (np.arange(-5000, 5000) + 0.5) / 100.0
    ll = np.concatenate([-rng.uniform(0, 20000, 100000), rng.normal(0, 1, 1000) * 1e6, ties2,
                         np.nextafter(ties2, 1e9), np.nextafter(ties2, -1e9),
                         [-0.004, -0.005, 0.005, 1e15, -1e300, np.nan, -np.inf, 199999.995, 200000.0]])
    assert mismatches(ll, 2) == []


def test_add_problem_rejects_fragment_lengths_outside_the_distribution():
    """A caller-made paired-end problem whose fragment length does not index the fragment-length
    distribution (mean 250, sd 30, 4 sd: 130..370) is an error, not an out-of-bounds table read."""
    match = np.ones((3, 2))
    for bad in (5, 100000, -1):
        b = miso_amd.Batch(36, iters=10, burn=2, lag=1, chains=1, paired=True, mean=250.0, var=900.0)
        fl = np.full((3, 2), 250, np.int32)
        fl[1, 0] = bad
        with pytest.raises(miso_amd.InternalError, match="Fragment length"):
            b.add_problem(match, [1000, 900], [3, 2], fraglen=fl)
    b = miso_amd.Batch(36, iters=10, burn=2, lag=1, chains=1, paired=True, mean=250.0, var=900.0)
    fl = np.full((3, 2), 250, np.int32)
    fl[1, 0] = -1
    m2 = match.copy(); m2[1, 0] = 0            # incompatible: its fragment length is not looked at
    assert b.add_problem(m2, [1000, 900], [3, 2], fraglen=fl) == 0


@pytest.mark.parametrize("K,paired", [(33, False), (64, False), (40, True), (65, False), (130, False), (256, False), (100, True), (200, True)])
def test_read_classes_of_genes_with_more_than_32_isoforms(orc, ref, K, paired):
    """Host packing with compatibility masks of several words (33 ... 256 isoforms; the reference has no limit, miso.c:696,
    gff.c:684): the read classes the header reports equal the REAL reference's (miso.c:762, miso_paired.c:386-391 through
    oracle/_ref), 257 isoforms are refused."""
    from _problems import se_gene, expr_for, flat
    exons, isoforms = se_gene(K, exlen=420 if paired else 60, gap=250 if paired else 50)
    g = ref.gene(flat(exons), isoforms)
    ref.rng_seed(31)
    if paired:
        rc, _, pos, cig = ref.simulate_paired_reads(g, expr_for(K), 300, 36, 250.0, 900.0)
        r = ref.miso_paired(g, pos, cig, 36, 250.0, 900.0, iters=10, burn=2, lag=1, chains=1)
    else:
        rc, _, pos, cig = ref.simulate_reads(g, expr_for(K), 500, 36)
        r = ref.miso(g, pos, cig, 36, iters=10, burn=2, lag=1, chains=1)
    assert rc == 0 and r.rc == 0
    b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0)
    i = b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    ct, cc = b.classes(i)
    assert np.array_equal(ct, r.class_templates) and np.array_equal(cc, r.class_counts)
    e257, i257 = se_gene(257, exlen=60, gap=50)
    with pytest.raises(NotImplementedError, match="More than 256 isoforms"):
        miso_amd.Batch(36).add_event(miso_amd.Gene(e257, i257), pos[:4], cig[:4])


def test_convergent_mean_rule_matches_the_checker_and_the_reference(orc):
    """The stopping rule of stop=CONVERGENT_MEAN (miso.c:556-636) as the library's host code computes it, against the
    checker's restatement and -- where oracle/_ref travelled -- the reference's own function: chains that agree, chains
    apart, few samples, constant samples (W = 0 -> NaN -> not converged)."""
    from _libs import RefLib
    ref = RefLib() if RefLib.available() else None
    rng = np.random.default_rng(5)
    cases = []
    for K, Cn, S, spread in [(2, 3, 60, 0.0), (2, 3, 60, 0.2), (3, 4, 160, 0.0), (5, 2, 40, 0.05), (2, 6, 2400, 0.0),
                             (4, 2, 4, 0.0), (2, 2, 2, 0.0), (10, 6, 120, 0.3)]:
        x = rng.dirichlet(np.ones(K) * 20, size=S)
        x += spread * (np.arange(S) % Cn)[:, None] * np.eye(K)[0]      # chain j shifted by j * spread in isoform 0
        cases.append((x, Cn))
    cases.append((np.full((30, 2), 0.5), 3))                            # constant
    seen = set()
    for x, Cn in cases:
        got = capi.selftest_convergent_mean(x, Cn)
        assert int(got) == orc.convergent_mean(x, Cn)
        if ref is not None:
            assert int(got) == ref.convergent_mean(x, Cn)
        seen.add(got)
    assert seen == {True, False}
    with pytest.raises(miso_amd.InternalError):
        capi.selftest_convergent_mean(np.zeros((1, 2)), 2)


def test_assignment_matrix_of_the_host_library(orc):
    """include/miso_amd.h miso_gene_assignment_matrix (what algorithm=CLASSES sums over; the module's assignmentMatrix)
    against the checker's, which is tested against the reference's own function (tests/test_oracle_vs_ref.py)."""
    from test_oracle_vs_ref import _random_genes
    from _problems import flat, se_gene
    for exons, isoforms in [se_gene(K) for K in (2, 3, 5, 8, 12, 40)] + _random_genes(np.random.default_rng(6), 30):
        G = miso_amd.Gene(exons, isoforms)
        og = orc.gene(flat(exons), isoforms)
        for read_len in (36, 75):
            assert np.array_equal(G.assignment_matrix(read_len), orc.assignment_matrix(og, read_len)), (exons, isoforms)
    with pytest.raises(NotImplementedError):
        G.assignment_matrix(36, overhang=2)
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miso_amd"))
    import pysplicing   # (the one module object tests/test_pysplicing_module.py and miso_sampler use)
    g = pysplicing.createGene(((1, 100), (201, 300), (401, 500)), ((0, 1, 2), (0, 2)))
    assert pysplicing.assignmentMatrix(g, 0, 36) == ((0.0, 135.0, 130.0), (35.0, 0.0, 130.0))
