"""The `pysplicing` drop-in module (miso_amd/pysplicing) and the miso_sampler.py mirror.

CPU part: API surface of the reference module (pysplicing.c:659-685, __init__.py:2-13), tuple-only
argument conversion (pyconvert.c), error mapping (pyerror.c), `.miso` file format
(miso_sampler.py:376-466; field order pinned by the reference's shipped .miso headers).
GPU part: MISO / MISOPaired / MISOBatch results equal the oracle's; run_sampler end to end."""
import os
import random
import sys

import numpy as np
import pytest

import _golden
from _libs import OrcLib
from _problems import flat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miso_amd"))
import pysplicing  # noqa: E402
from miso_amd import miso_sampler  # noqa: E402

# header fields of misopy/sashimi_plot/test-data/miso-data/heartWT1/chr17/*.miso, in order
REF_HEADER_FIELDS = ["isoforms", "exon_lens", "iters", "burn_in", "lag", "percent_accept",
                     "proposal_type", "counts", "assigned_counts", "chrom", "strand", "mRNA_starts",
                     "mRNA_ends"]


def test_module_surface():
    names = ["readGFF", "writeGFF", "createGene", "simulateReads", "assignmentMatrix", "noIso",
             "isoLength", "solveIsoGene", "simulatePairedReads", "MISO", "MISOPaired",
             "geneComplexity", "noGenes", "i_fromGFF", "toGFF", "InternalError"]
    for n in names:
        assert hasattr(pysplicing, n), n
    assert (pysplicing.MISO_START_AUTO, pysplicing.MISO_START_UNIFORM, pysplicing.MISO_START_RANDOM,
            pysplicing.MISO_START_GIVEN, pysplicing.MISO_START_LINEAR) == (0, 1, 2, 3, 4)
    assert (pysplicing.MISO_STOP_FIXEDNO, pysplicing.MISO_STOP_CONVERGENT_MEAN) == (0, 1)
    assert (pysplicing.MISO_ALGO_REASSIGN, pysplicing.MISO_ALGO_MARGINAL, pysplicing.MISO_ALGO_CLASSES) == (0, 1, 2)
    for off_path in ("readGFF", "solveIsoGene", "geneComplexity"):
        with pytest.raises(NotImplementedError):
            getattr(pysplicing, off_path)("x")


def test_tuple_only_arguments_and_errors():
    g = pysplicing.createGene(((1, 100), (201, 300), (401, 500)), ((0, 1, 2), (0, 2)))
    assert pysplicing.noIso(g) == (2,) and pysplicing.isoLength(g) == ((300, 200),)
    with pytest.raises(TypeError):
        pysplicing.createGene([(1, 100)], ((0,),))             # list rejected: pyconvert.c:41-44
    with pytest.raises(TypeError):
        pysplicing.MISO(g, 0, [10], ("36M",), 36)              # pyconvert.c:7-10
    with pytest.raises(TypeError):
        pysplicing.MISO("not a gene", 0, (10,), ("36M",), 36)
    with pytest.raises(pysplicing.InternalError, match="Overhang length invalid"):
        pysplicing.MISO(g, 0, (10,), ("36M",), 36, 100, 10, 1, (1.0, 1.0), 20)
    with pytest.raises(pysplicing.InternalError, match="Unsupported CIGAR"):
        pysplicing.MISO(g, 0, (10,), ("36Q",), 36, 100, 10, 1, (1.0, 1.0), 1, 1)
    with pytest.raises(NotImplementedError):
        pysplicing.MISO(g, 0, (10,), ("36M",), 36, 100, 10, 1, (1.0, 1.0), 1, 2, 2)      # START_RANDOM
    random.seed(5)
    a = pysplicing.simulateReads(g, 0, (0.3, 0.7), 20, 36)
    random.seed(5)
    b = pysplicing.simulateReads(g, 0, (0.3, 0.7), 20, 36)
    assert a == b and len(a) == 3 and len(a[1]) == 20


def test_miso_file_format(tmp_path):
    gene = miso_sampler.SimpleGene([(1, 91), (201, 239), (401, 480)], [[0, 1, 2], [0, 2]],
                                   label="ev", chrom="chr17", strand="-")
    s = miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    psi = np.array([[0.12341, 0.87659], [0.5, 0.5]])
    out = str(tmp_path / "chr17" / "ev.miso")
    s.output_miso_results(out, gene, (((0.0, 1.0), (1.0, 0.0), (1.0, 1.0)), (1.0, 21.0, 23.0)),
                          np.array([0] * 34 + [1] * 11 + [-1] * 2), psi, np.array([-989.1296, -1002.117]),
                          2000, 200, 5, 95.2, "drift")
    lines = open(out).read().split("\n")
    fields = [kv.split("=", 1)[0] for kv in lines[0][1:].split("\t")]
    assert fields == REF_HEADER_FIELDS
    assert "counts=(0,1):1,(1,0):21,(1,1):23\tassigned_counts=0:34,1:11\tchrom=chr17\tstrand=-" in lines[0]
    assert "iters=2000\tburn_in=200\tlag=5\tpercent_accept=95.20\tproposal_type=drift" in lines[0]
    assert "exon_lens=('ev.0',91),('ev.1',39),('ev.2',80)" in lines[0]
    assert lines[1] == "sampled_psi\tlog_score"
    assert lines[2] == "0.1234,0.8766\t-989.13" and lines[3] == "0.5000,0.5000\t-1002.12"
    samples, hdr, scores = miso_sampler.load_samples(out)
    assert samples.shape == (2, 2) and hdr["lag"] == "5" and scores[0] == -989.13


def test_skip_rules(tmp_path, capsys):
    s = miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    gene = miso_sampler.SimpleGene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]])
    assert s.run_sampler(100, ((), ()), gene, None, s.params, str(tmp_path / "a")) is None     # no reads
    (tmp_path / "b.miso").write_text("x")
    assert s.run_sampler(100, ((5,), ("36M",)), gene, None, s.params, str(tmp_path / "b")) is None
    one = miso_sampler.SimpleGene([(1, 100)], [[0]])
    assert s.run_sampler(100, ((5,), ("36M",)), one, None, s.params, str(tmp_path / "c")) is None


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_miso_matches_oracle_and_python_random_seeding(orc):
    g = _golden.load("se_k3")
    cg = pysplicing.createGene(tuple(g["exon_list"]), tuple(tuple(i) for i in g["isoform_list"]))
    pos = tuple(int(p) for p in g["pos"])
    cig = tuple(c.decode() for c in g["cigars"])
    args = (cg, 0, pos, cig, g["read_len"], g["iters"], g["burn"], g["lag"], (1.0, 1.0, 1.0),
            g["overhang"], g["chains"])
    res = pysplicing.MISO(*args, seed=123)
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"],
                   lag=g["lag"], chains=g["chains"], overhang=g["overhang"], mode=OrcLib.COUNTER,
                   seed=123, event_id=0)
    assert np.array_equal(np.array(res[0]).T, cpu.samples)
    assert np.array_equal(np.array(res[1]), cpu.loglik)
    assert np.array_equal(np.array(res[2]), g["class_templates"])
    assert np.array_equal(np.array(res[3]), g["class_counts"])
    assert np.array_equal(np.array(res[4]), cpu.assignment)
    assert res[5] == (3, g["iters"], g["burn"], g["lag"], cpu.accepted, cpu.rejected)
    random.seed(77)
    a = pysplicing.MISO(*args)
    random.seed(77)
    b = pysplicing.MISO(*args)
    c = pysplicing.MISO(*args)
    assert a == b and a[0] != c[0]


@pytest.mark.gpu
def test_batch_and_paired_entry_points(orc):
    gs = [_golden.load(n) for n in ("se_k2", "se_k5", "cigar_edges")]
    genes = [pysplicing.createGene(tuple(g["exon_list"]), tuple(tuple(i) for i in g["isoform_list"]))
             for g in gs]
    events = tuple((cg, tuple(int(p) for p in g["pos"]), tuple(c.decode() for c in g["cigars"]))
                   for cg, g in zip(genes, gs))
    out = pysplicing.MISOBatch(events, 36, 400, 100, 3, 1, 2, seed=9, first_event_id=50)
    assert len(out) == 3
    for i, g in enumerate(gs):
        og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
        cpu = orc.miso(og, g["pos"], g["cigars"], 36, iters=400, burn=100, lag=3, chains=2,
                       mode=OrcLib.COUNTER, seed=9, event_id=50 + i)
        assert np.array_equal(np.array(out[i][0]).T, cpu.samples), i
        assert np.array_equal(np.array(out[i][4]), cpu.assignment), i
    p = _golden.load("pe_k2")
    cg = pysplicing.createGene(tuple(p["exon_list"]), tuple(tuple(i) for i in p["isoform_list"]))
    res = pysplicing.MISOPaired(cg, 0, tuple(int(v) for v in p["pos"]),
                                tuple(c.decode() for c in p["cigars"]), 36, 250.0, 900.0, 4.0, 300, 50,
                                2, (1.0, 1.0), 1, 2, seed=4)
    og = orc.gene(flat(p["exon_list"]), p["isoform_list"])
    cpu = orc.miso_paired(og, p["pos"], p["cigars"], 36, 250.0, 900.0, iters=300, burn=50, lag=2,
                          chains=2, mode=OrcLib.COUNTER, seed=4, event_id=0)
    assert np.array_equal(np.array(res[0]).T, cpu.samples)
    assert np.array_equal(np.array(res[2]), p["class_templates"])      # binary classes, as the reference


@pytest.mark.gpu
def test_run_sampler_end_to_end_on_reference_test_data(tmp_path, orc):
    """config[0]: the reference's own test gene (Atp2b1, 2 isoforms, 3589 single-end reads)."""
    g = _golden.load("atp2b1")
    gene = miso_sampler.SimpleGene(g["exon_list"], g["isoform_list"], label="ENSMUSG00000019943",
                                   chrom="10", strand="+")
    s = miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    reads = (tuple(int(p) - 1 for p in g["pos"]), tuple(g["cigars"]))     # 0-based, as sam_utils gives
    out = s.run_sampler(1000, reads, gene, None, s.params, str(tmp_path / "10" / "atp2b1"),
                        num_chains=2, burn_in=200, lag=4, verbose=False, seed=31)
    samples, hdr, scores = miso_sampler.load_samples(out)
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    cpu = orc.miso(og, g["pos"], g["cigars"], 36, iters=1000, burn=200, lag=4, chains=2,
                   mode=OrcLib.COUNTER, seed=31, event_id=0)
    assert np.array_equal(samples, np.round(cpu.samples, 4))
    counts = ",".join("(%s):%d" % (",".join(str(int(v)) for v in t), c)
                      for t, c in zip(g["class_templates"], g["class_counts"]))
    assert hdr["counts"] == counts and hdr["iters"] == "1000" and hdr["chrom"] == "10"
    # posterior mean within Monte-Carlo error of the REAL reference's run stored in the fixture
    assert abs(samples[:, 0].mean() - g["samples"][:, 0].mean()) < 0.02
    ev = [((reads[0][:500], reads[1][:500]), gene, str(tmp_path / "b" / ("e%d" % i))) for i in range(3)]
    ev.append((((), ()), gene, str(tmp_path / "b" / "empty")))
    written = s.run_sampler_batch(300, ev, num_chains=1, burn_in=50, lag=1, seed=2)
    assert written[3] is None and all(w and os.path.exists(w) for w in written[:3])
    a, b = (miso_sampler.load_samples(w)[0] for w in written[:2])
    assert not np.array_equal(a, b)        # different event ids -> different streams


@pytest.mark.gpu
def test_legacy_test_pysplicing_script_shape():
    """misopy/legacy_test_pysplicing.py:9-21, call for call (3 exons, isoforms (0,1),(0,2),(0,1,2),
    expression (.2,.3,.5), 2000 reads of 33 bp, MISO(…, 5000, 500, 10, (1,1,1)))."""
    gene = pysplicing.createGene(((1, 100), (201, 300), (401, 500)), ((0, 1), (0, 2), (0, 1, 2)))
    assert pysplicing.noIso(gene) == (3,)                      # one entry per gene of the handle
    assert pysplicing.isoLength(gene) == ((200, 200, 300),)
    reads = pysplicing.simulateReads(gene, 0, (0.2, 0.3, 0.5), 2000, 33, seed=7)
    assert len(reads[1]) == len(reads[2]) == 2000 and isinstance(reads[1], tuple)
    results = pysplicing.MISO(gene, 0, reads[1], reads[2], 33, 5000, 500, 10, (1.0, 1.0, 1.0), seed=11)
    samples, loglik, templates, counts, assignment, rundata = results
    S = 6 * (5000 - 500) // 10                       # default 6 chains (pysplicing.c:62-66)
    assert len(samples) == 3 and all(len(row) == S for row in samples) and len(loglik) == S
    assert len(assignment) == 2000 and rundata[:4] == (3, 5000, 500, 10)
    assert sum(counts) == 2000
    psi = np.array(samples).mean(axis=1)
    assert abs(psi.sum() - 1) < 1e-9
    assert np.all(np.abs(psi - np.array([0.2, 0.3, 0.5])) < 0.06)


def test_batch_path_skips_a_bad_gene_and_the_single_event_path_raises(tmp_path, capsys):
    """A bad CIGAR: the one-event call raises pysplicing.InternalError as the reference's does (pyerror.c:27-44); inside
    a batch the gene is reported and skipped -- the reference's worker runs each gene in its own try block
    (run_miso.py:205-256) -- and nothing is written for it."""
    gene = miso_sampler.SimpleGene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]])
    s = miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    written = s.run_sampler_batch(100, [(((10, 20), ("36M", "3Q")), gene, str(tmp_path / "x"))],
                                  num_chains=1, burn_in=10, lag=1, seed=1)
    assert written == [None] and not os.path.exists(str(tmp_path / "x.miso"))
    assert len(s.skipped_genes) == 1 and "CIGAR" in s.skipped_genes[0][1]
    assert "Skipping gene" in capsys.readouterr().out
    cg = pysplicing.createGene(((1, 100), (201, 300), (401, 500)), ((0, 1, 2), (0, 2)))
    with pytest.raises(pysplicing.InternalError, match="CIGAR"):
        pysplicing.MISO(cg, 0, (11, 21), ("36M", "3Q"), 36, 100, 10, 1, (1.0, 1.0), 1, 1)
