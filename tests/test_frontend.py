"""CPU: the front end -- GFF3 -> genes, SAM/BAM -> reads of an event, the
indexer and the batch collector -- against the reference's own test data set
(tests/golden/data: misopy/gff-events/mm9/genes/Atp2b1.mm9.gff and
misopy/test-data/sam-data/c2c12.Atp2b1.sam, the inputs of misopy/test_miso.py:131-171), the
golden arrays extracted from it (tests/golden/atp2b1.npz) and brute-force restatements."""
import gzip
import os
import random

import numpy as np
import pytest

from _bam import sam_to_bam
from _golden import load as load_golden

from miso_amd import gene_utils, gff_utils, index_gff, run_miso, sam_utils
from miso_amd.settings import Settings

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
GFF = os.path.join(DATA, "Atp2b1.mm9.gff")


@pytest.fixture(scope="module")
def sam_text():
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        return f.read()


@pytest.fixture(scope="module")
def sam_path(tmp_path_factory, sam_text):
    p = tmp_path_factory.mktemp("aln") / "c2c12.Atp2b1.sam"
    p.write_text(sam_text)
    return str(p)


@pytest.fixture(scope="module")
def bam_path(tmp_path_factory, sam_text):
    p = str(tmp_path_factory.mktemp("aln") / "c2c12.Atp2b1.bam")
    sam_to_bam(sam_text, p, block=20000)     # ~60 BGZF blocks: the parallel inflate path
    return p


# ---- GFF -> gene ------------------------------------------------------------------------------
def test_gene_from_gff_matches_golden_structure():
    g = load_golden("atp2b1")
    genes = gene_utils.load_genes_from_gff(GFF, suppress_warnings=True)
    assert list(genes) == ["ENSMUSG00000019943"]
    gene = genes["ENSMUSG00000019943"]["gene_object"]
    assert (gene.chrom, gene.strand) == ("10", "+")
    assert [iso.label for iso in gene.isoforms] == list(g["mrna_ids"])
    for iso, idx in zip(gene.isoforms, g["isoform_list"]):
        want = [g["exon_list"][i] for i in idx]
        assert [(p.start, p.end) for p in iso.parts] == want
    # parts = every exon of every transcript, shared exons once per transcript (Gene.py:992-1000)
    assert len(gene.parts) == sum(len(iso.parts) for iso in gene.isoforms)
    assert list(gene.iso_lens) == [1364, 4755]
    tx = gff_utils.get_inclusive_txn_bounds(genes["ENSMUSG00000019943"]["hierarchy"]["ENSMUSG00000019943"])
    assert tx == (98377804, 98486420)


def test_py2c_gene_resolves_shared_exons_to_the_first_equal_part():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(DATA), "..", "..", "miso_amd"))
    genes = gene_utils.load_genes_from_gff(GFF, suppress_warnings=True)
    gene = genes["ENSMUSG00000019943"]["gene_object"]
    # py2c_gene.py:10-21 semantics without the extension: indices via list.index / Exon.__eq__
    iso2 = [gene.parts.index(p) for p in gene.isoforms[1].parts]
    shared = [(p.start, p.end) for p in gene.isoforms[0].parts]
    for i, p in zip(iso2, gene.isoforms[1].parts):
        if (p.start, p.end) in shared:
            assert i < len(gene.isoforms[0].parts)       # points into transcript 1's copy
        assert (gene.parts[i].start, gene.parts[i].end) == (p.start, p.end)


def test_gff_reader_details(tmp_path):
    p = tmp_path / "t.gff"
    p.write_text("##gff-version 3\n# comment\n\n"
                 "chr1\tsrc\tgene\t100\t900\t.\t-\t.\tID=g%201;Name=a,b\n"
                 "chr1\tsrc\tmRNA\t900\t100\t.\t-\t.\tID=t1;Parent=g%201\n"          # swapped coords
                 "chr1\tsrc\texon\t100\t200\t.\t-\t.\tParent=t1\n"                     # default exon ID
                 "chr1\tsrc\texon\t300\t400\t.\t-\t.\tID=e2;Parent=t1;bad\n")
    db = gff_utils.GFFDatabase(str(p), suppress_warnings=True)
    assert db.genes[0].get_id() == "g 1" and db.genes[0].attributes["Name"] == ["a", "b"]
    assert (db.mRNAs[0].start, db.mRNAs[0].end) == (100, 900)
    assert db.exons[0].get_id() == "t1@100@200@-"
    genes = gene_utils.load_genes_from_gff(str(p), suppress_warnings=True)
    gene = genes["g 1"]["gene_object"]
    assert [p_.label for p_ in gene.parts] == ["t1@100@200@-", "e2"] and gene.strand == "-"
    with pytest.raises(gff_utils.FormatError):
        list(gff_utils.Reader(iter(["a\tb\tc\n"])))


# ---- SAM / BAM reader -------------------------------------------------------------------------
def test_sam_reader_matches_golden_arrays(sam_path):
    g = load_golden("atp2b1")
    f = sam_utils.Samfile(sam_path)
    assert not f.is_bam and "10" in f.references and f.lengths[f.gettid("10")] == 129993255
    (pos, cig), n = f.parse_reads("10", 0, 1 << 30)
    assert n == len(g["pos"]) == 3589
    assert np.array_equal(np.asarray(pos) + 1, g["pos"])            # golden positions are 1-based
    assert list(cig) == [c.decode() for c in g["cigars"]]


def test_bam_equals_sam(sam_path, bam_path):
    a, b = sam_utils.Samfile(sam_path), sam_utils.Samfile(bam_path, threads=4)
    assert b.is_bam and a.references == b.references and a.lengths == b.lengths
    for col in ("ref_id", "pos", "end", "flag", "l_seq", "cigar_off", "cigar", "name_off"):
        assert np.array_equal(getattr(a, col), getattr(b, col)), col
    assert a.names == b.names
    r = b.read(5)
    assert r.cigar is not None and sam_utils.sam_cigar_to_str(r.cigar) and r.rlen == 36


def test_fetch_equals_brute_force(bam_path):
    f = sam_utils.Samfile(bam_path)
    tid = f.gettid("10")
    rng = random.Random(3)
    lo, hi = int(f.pos.min()), int(f.end.max())
    for _ in range(200):
        s = rng.randint(lo - 1000, hi + 1000)
        e = s + rng.choice([0, 1, 50, 5000, 200000])
        want = np.nonzero((f.ref_id == tid) & (f.pos < e) & (f.end > s))[0]
        got = f.fetch_indices("10", s, e)
        assert sorted(got.tolist()) == want.tolist()
        assert np.all(np.diff(f.pos[got]) >= 0)
    with pytest.raises(ValueError):
        f.fetch("nope", 0, 10)
    assert sam_utils.fetch_bam_reads_in_gene(f, "chr10", lo, lo + 100) != []    # 'chr' stripped
    assert sam_utils.fetch_bam_reads_in_gene(f, "chrZ", 0, 10) == []


def test_unsorted_input_and_corrupt_files(tmp_path, sam_text):
    lines = sam_text.splitlines()
    head = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    random.Random(1).shuffle(body)
    p = tmp_path / "shuffled.bam"
    sam_to_bam("\n".join(head + body) + "\n", str(p), block=7000)
    f = sam_utils.Samfile(str(p))
    (pos, cig), n = f.parse_reads("10", 98377804, 98486420)
    assert n > 3000 and list(pos) == sorted(pos)
    raw = p.read_bytes()
    bad = tmp_path / "bad.bam"
    bad.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(IOError):
        sam_utils.Samfile(str(bad))
    bad.write_bytes(raw[:100] + bytes(50) + raw[150:])
    with pytest.raises(IOError):
        sam_utils.Samfile(str(bad))
    with pytest.raises(IOError):
        sam_utils.Samfile(str(tmp_path / "missing.bam"))
    bad.write_text("r1\t0\tchr1\t5\n")
    with pytest.raises(IOError):
        sam_utils.Samfile(str(bad))
    empty = tmp_path / "empty.sam"
    empty.write_text("@SQ\tSN:chr1\tLN:100\n")
    e = sam_utils.Samfile(str(empty))
    assert len(e) == 0 and e.parse_reads("chr1", 0, 100) == (((), ()), 0)


# ---- read filters, pairing, strand rules: native path vs the line-by-line mirror ---------------
def synthetic_pairs(seed, n=300):
    rng = random.Random(seed)
    lines = ["@SQ\tSN:chr1\tLN:100000"]
    for i in range(n):
        name = "r%d" % i
        p1 = rng.randint(100, 5000)
        p2 = p1 + rng.randint(0, 400)
        rev1 = rng.random() < 0.5
        rev2 = (not rev1) if rng.random() < 0.9 else rev1                 # some same-strand pairs
        first_is_1 = rng.random() < 0.5
        f1 = 0x1 | (0x10 if rev1 else 0) | (0x20 if rev2 else 0) | (0x40 if first_is_1 else 0x80)
        f2 = 0x1 | (0x10 if rev2 else 0) | (0x20 if rev1 else 0) | (0x80 if first_is_1 else 0x40)
        kind = rng.random()
        if kind < 0.05:
            f1 |= 0x200                                                   # QC fail
        elif kind < 0.10:
            f2 |= 0x8                                                     # mate unmapped
        elif kind < 0.15:
            f1 &= ~0x1                                                    # not paired
        suffix = rng.choice([("", ""), ("/1", "/2"), ("#1", "#2")])
        l1 = 36 if rng.random() < 0.9 else 30
        cig1 = "%dM" % l1 if rng.random() < 0.8 else "10M200N%dM" % (l1 - 10)
        cig2 = "36M" if rng.random() < 0.95 else "*"
        recs = [(name + suffix[0], f1, p1, cig1, l1), (name + suffix[1], f2, p2, cig2, 36)]
        if kind > 0.95:
            recs.append((name + suffix[0], f1, p1 + 3, cig1, l1))         # a third alignment
        if kind > 0.90 and kind <= 0.95:
            recs = recs[:1]                                               # mate missing
        for nm, fl, ps, cg, ln in recs:
            lines.append("\t".join([nm, str(fl), "chr1", str(ps), "255", cg, "=", "1", "0",
                                    "A" * ln, "I" * ln]))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("rule,target", [(None, None), ("fr-unstranded", "+"), ("fr-firststrand", "+"),
                                         ("fr-firststrand", "-"), ("fr-firststrand", None),
                                         ("fr-firststrand", "?")])
@pytest.mark.parametrize("rlen", [None, 36])
def test_native_parse_reads_equals_reference_logic(tmp_path, paired, rule, target, rlen):
    p = tmp_path / "pairs.sam"
    p.write_text(synthetic_pairs(11))
    f = sam_utils.Samfile(str(p))
    paired = (250, 30) if paired else None       # what run_miso.py passes: None or (mean, sd)
    for (s, e) in [(4990, 5001), (1000, 3000), (0, 100000)]:
        native, n_native = f.parse_reads("chr1", s, e, paired_end=paired, strand_rule=rule,
                                         target_strand=target, given_read_len=rlen)
        mirror, n_mirror = sam_utils.sam_parse_reads(f.fetch("chr1", s, e), paired_end=paired,
                                                     strand_rule=rule, target_strand=target,
                                                     given_read_len=rlen)
        assert n_native == n_mirror
        assert native == mirror
    assert n_native > 0 or target == "?"


def test_strand_rules_of_the_reference_test_suite():
    """misopy/test_miso.py:80-127 (test_strandedness), same assertions."""
    f_read = sam_utils.AlignedRead("f", 0, 0, [(0, 36)], 36)
    r_read = sam_utils.AlignedRead("r", 16, 0, [(0, 36)], 36)
    m = sam_utils.read_matches_strand
    for read in (f_read, r_read):
        for target in "+-":
            assert m(read, target, "fr-unstranded") is True
    assert m(f_read, "+", "fr-firststrand") is True
    assert m(f_read, "-", "fr-firststrand") is False
    assert m(r_read, "+", "fr-firststrand") is False
    assert m(r_read, "-", "fr-firststrand") is True
    pe = (300, 10)
    assert m((f_read, r_read), "+", "fr-firststrand", paired_end=pe) is True
    assert m((f_read, r_read), "-", "fr-firststrand", paired_end=pe) is True
    with pytest.raises(Exception):
        m(f_read, "+", "fr-secondstrand")
    assert sam_utils.strip_mate_id("read7/1") == "read" and sam_utils.strip_mate_id("read7") == "read7"
    assert sam_utils.flag_to_strand(16) == "-" and sam_utils.flag_to_strand(99) == "+"


# ---- index + collector ---------------------------------------------------------------------------
def test_index_and_collect(tmp_path, bam_path):
    idx = str(tmp_path / "indexed")
    index_gff.index_gff(GFF, idx)
    pick = os.path.join(idx, "chr10", "ENSMUSG00000019943.pickle")
    assert os.path.isfile(pick) and os.path.isfile(os.path.join(idx, "genes.gff"))
    m = gff_utils.get_gene_ids_to_gff_index(idx)
    assert list(m.items()) == [("ENSMUSG00000019943", pick)]
    os.remove(os.path.join(idx, gff_utils.INDEX_MAP_BASENAME))          # directory-scan fallback
    assert dict(gff_utils.get_gene_ids_to_gff_index(idx)) == dict(m)
    index_gff.index_gff(GFF, idx)                                         # already indexed: no-op
    Settings.load(None)
    bam = sam_utils.Samfile(bam_path)
    out = str(tmp_path / "out")
    events, info = run_miso.collect_gene_events(m.items(), bam, out, 36, 1, verbose=False)
    assert len(events) == 1 and info["ENSMUSG00000019943"].endswith("reads")
    (pos, cig), gene, fname = events[0][:3]
    assert events[0][3] is None and events[0][4] == 0          # no prior; its number in the gene list
    assert fname == os.path.join(out, "10", "ENSMUSG00000019943") and len(pos) == len(cig) > 3000
    assert gene.label == "ENSMUSG00000019943"
    # the read-length and minimum-read filters of run_miso.py:110-115, 139-147
    assert run_miso.collect_gene_events(m.items(), bam, out, 5000, 1, verbose=False)[0] == []
    ev, info = run_miso.collect_gene_events(m.items(), bam, out, 40, 1, verbose=False)
    assert ev == [] and info["ENSMUSG00000019943"] == "only 0 reads"
    c = index_gff.compress_event_name("ENSMUSG00000019943")
    assert c.startswith("misocomp_") and c == index_gff.compress_event_name("ENSMUSG00000019943")


def test_settings(tmp_path):
    Settings.load(None)
    assert Settings.get_sampler_params() == {"num_chains": 6, "burn_in": 500, "lag": 10, "num_iters": 5000}
    assert Settings.get_min_event_reads() == 20 and Settings.get_strand_param() == "fr-unstranded"
    p = tmp_path / "s.txt"
    p.write_text("[data]\nfilter_results = True\nmin_event_reads = 5\nstrand = fr-firststrand\n"
                 "[cluster]\ncluster_command = long\n[sampler]\nburn_in = 10\nlag = 2\nnum_iters = 100\n")
    Settings.load(str(p))
    assert Settings.get_sampler_params() == {"num_chains": 6, "burn_in": 10, "lag": 2, "num_iters": 100}
    assert Settings.get_min_event_reads() == 5 and Settings.get_strand_param() == "fr-firststrand"
    assert Settings.get()["cluster_command"] == "long"
    Settings.load(None)


def test_dispatcher_chunks_like_the_reference(tmp_path, bam_path, monkeypatch):
    """miso.py:152-186 / cluster_utils.py:23-32: contiguous chunks of the gene list, one per GPU,
    each knowing the global index of its first event (results independent of the split)."""
    from miso_amd import miso as miso_cli
    idx = str(tmp_path / "indexed")
    genes = tmp_path / "many.gff"
    lines = ["##gff-version 3"]
    for g in range(11):
        s = 1000 + 10000 * g
        lines += ["chr1\tx\tgene\t%d\t%d\t.\t+\t.\tID=g%02d" % (s, s + 900, g),
                  "chr1\tx\tmRNA\t%d\t%d\t.\t+\t.\tID=g%02d.A;Parent=g%02d" % (s, s + 900, g, g),
                  "chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=g%02d.A.1;Parent=g%02d.A" % (s, s + 100, g, g),
                  "chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=g%02d.A.2;Parent=g%02d.A" % (s + 800, s + 900, g, g),
                  "chr1\tx\tmRNA\t%d\t%d\t.\t+\t.\tID=g%02d.B;Parent=g%02d" % (s, s + 900, g, g),
                  "chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=g%02d.B.1;Parent=g%02d.B" % (s, s + 100, g, g),
                  "chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=g%02d.B.2;Parent=g%02d.B" % (s + 400, s + 500, g, g),
                  "chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=g%02d.B.3;Parent=g%02d.B" % (s + 800, s + 900, g, g)]
    genes.write_text("\n".join(lines) + "\n")
    index_gff.index_gff(str(genes), idx)
    assert miso_cli.chunk_list(list(range(11)), 4) == [[0, 1], [2, 3, 4], [5, 6, 7], [8, 9, 10]]
    assert miso_cli.chunk_list(list(range(3)), 1) == [[0, 1, 2]]
    d = miso_cli.GenesDispatcher(idx, bam_path, str(tmp_path / "out"), 36, 1, num_proc=4, seed=3)
    batches = d.output_batch_files()
    assert [(n, first) for _, n, first in batches] == [(2, 0), (3, 2), (3, 5), (3, 8)]
    seen = []
    for fname, n, first in batches:
        entries = run_miso.read_genes_file(fname)
        assert len(entries) == n and all(os.path.isfile(p) for _, p in entries)
        seen += [g for g, _ in entries]
    assert seen == ["g%02d" % g for g in range(11)]
    with pytest.raises(IOError):
        miso_cli.GenesDispatcher(idx, str(tmp_path / "no.bam"), str(tmp_path / "o2"), 36, 1, num_proc=1)


def test_parallel_sam_parse_equals_serial(tmp_path, sam_text):
    lines = sam_text.splitlines()
    head = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    big = tmp_path / "big.sam"
    with open(big, "w") as f:
        f.write("\n".join(head) + "\n")
        for rep in range(40):                                 # ~22 MB: several 4 MB chunks per thread
            f.write("\n".join(body) + "\n")
    a = sam_utils.Samfile(str(big), threads=1)
    b = sam_utils.Samfile(str(big), threads=5)
    assert len(a) == len(b) == 40 * len(body)
    for col in ("ref_id", "pos", "end", "flag", "l_seq", "cigar_off", "cigar", "name_off"):
        assert np.array_equal(getattr(a, col), getattr(b, col)), col
    assert a.names == b.names and a.references == b.references
    # a reference the header does not list: the parallel path hands over to the serial one
    with open(big, "a") as f:
        f.write("late\t0\tchrUn\t5\t255\t36M\t*\t0\t0\t%s\t%s\n" % ("A" * 36, "I" * 36))
    c = sam_utils.Samfile(str(big), threads=5)
    assert len(c) == len(a) + 1 and c.references[-1] == "chrUn"
    assert c.fetch("chrUn", 0, 100)[0].qname == "late"


def test_batched_add_from_alignment_file_equals_single_adds(bam_path):
    """miso_batch_add_events_aln (threads) against a loop of miso_batch_add_event_aln: same events in
    the same order, same read counts, skipped regions reported the same way; a bad gene fails the call
    and adds nothing."""
    from miso_amd import capi
    bam = sam_utils.Samfile(bam_path)
    gene = gene_utils.load_genes_from_gff(GFF, suppress_warnings=True)["ENSMUSG00000019943"]["gene_object"]
    exons = [(p.start, p.end) for p in gene.parts]
    isoforms = [[gene.parts.index(p) for p in iso.parts] for iso in gene.isoforms]
    g = capi.Gene(exons, isoforms)
    regions = [("10", 98377804, 98486420), ("10", 5, 10), ("10", 98431328, 98457192),
               ("10", 98377804, 98378005)] * 5
    one = capi.Batch(36, iters=100, burn=10, lag=1, chains=1, device_match=True)
    single = [one.add_event_aln(g, bam, c, s, e, 0, "+", 36, 20) for c, s, e in regions]
    many = capi.Batch(36, iters=100, burn=10, lag=1, chains=1, device_match=True)
    idx, cnt = many.add_events_aln([g] * len(regions), bam, [r[0] for r in regions], [r[1] for r in regions],
                                   [r[2] for r in regions], 0, ["+"] * len(regions), 36, 20, threads=4)
    assert [(int(i), int(n)) for i, n in zip(idx, cnt)] == single
    assert len(one) == len(many) == sum(1 for i, _ in single if i >= 0) and len(many) >= 10
    assert single[1] == (-1, 0)
    lone = capi.Gene([(1, 100)], [[0]])                       # one isoform: refused like add_event
    with pytest.raises(capi.InternalError, match="two isoforms"):
        many.add_events_aln([g, lone], bam, ["10", "10"], [98377804] * 2, [98486420] * 2, 0, ["+", "+"], 36, 20)
    assert len(many) == len(one)


def test_reader_only_library_and_fork_inheritance(bam_path):
    """`miso --run`'s dispatcher (miso_amd/miso.py): the alignment file decoded ONCE through
    libmiso_aln.so -- the reader without the HIP kernels, so the parent never initialises the GPU
    runtime -- and inherited by a forked worker, whose load_bam_reads() returns the parent's object;
    the worker's fetch / parse_reads give what a fresh decode through the full library gives."""
    import multiprocessing
    import subprocess
    import sys
    code = r"""
import os, sys, multiprocessing
import numpy as np
sys.path.insert(0, %r)
from miso_amd import sam_utils
sam_utils.use_reader_library()
path = %r
sam_utils._PRELOADED[path] = sam_utils.Samfile(path, "rb")
assert sam_utils._native()._name.endswith("libmiso_aln.so")
assert not any("libamdhip64" in l or "libmiso_amd.so" in l for l in open("/proc/self/maps"))   # no HIP in the parent
def child(q):
    f = sam_utils.load_bam_reads(path)
    assert f is sam_utils._PRELOADED[path]
    tid = f.references[0]
    (pos, cig), n = f.parse_reads(tid, 0, 10 ** 9)
    q.put((f.mapped_plus_unmapped, len(pos), tuple(cig[:3]), n))
ctx = multiprocessing.get_context("fork")
q = ctx.Queue()
p = ctx.Process(target=child, args=(q,)); p.start()
got = q.get(timeout=60); p.join()
print(repr(got))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code % (root, os.path.abspath(bam_path))], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    fresh = sam_utils.Samfile(os.path.abspath(bam_path), "rb")
    (pos, cig), n = fresh.parse_reads(fresh.references[0], 0, 10 ** 9)
    want = (fresh.mapped_plus_unmapped, len(pos), tuple(cig[:3]), n)
    assert eval(out.stdout.strip().splitlines()[-1]) == want


def test_one_bad_gene_does_not_cost_the_batch_its_other_genes(bam_path, capsys):
    """run_miso.py:205-256 runs every gene in its own try block; prepare_batch must do the same inside one GPU batch:
    a gene with more isoforms than the kernels hold (258 > MISO_MAX_ISOFORMS = 256), a gene whose reads carry a malformed CIGAR and a gene that
    fails inside the batched native add are reported and skipped, every other gene stays in the batch."""
    import miso_sampler as miso
    from miso_sampler import AlnRegion, SimpleGene
    params = miso.get_single_end_sampler_params(2, 36, 1)
    sampler = miso.MISOSampler(params, paired_end=False)
    good = SimpleGene([(1, 100), (201, 300), (401, 500)], [[0, 1, 2], [0, 2]], label="good")
    wide_exons = [(1 + 200 * i, 100 + 200 * i) for i in range(260)]
    too_many = SimpleGene(wide_exons, [[0, 259]] + [[0, k, 259] for k in range(1, 258)], label="sixty-six-isoforms")
    ok_reads = ([10, 220, 60], ["36M", "36M", "36M"])
    bad_reads = ([10, 220], ["36M", "3x6M"])
    gene_obj = gene_utils.load_genes_from_gff(GFF, suppress_warnings=True)["ENSMUSG00000019943"]["gene_object"]
    bam = sam_utils.Samfile(bam_path)
    region = AlnRegion(bam, "10", 98377804, 98486420, "fr-unstranded", "+", 36, 20)
    events = [(ok_reads, good, "/tmp/_miso_t/good1"), (ok_reads, too_many, "/tmp/_miso_t/many"),
              (bad_reads, good, "/tmp/_miso_t/badcigar"), (region, gene_obj, "/tmp/_miso_t/aln1"),
              (region, too_many, "/tmp/_miso_t/aln_many"), (region, gene_obj, "/tmp/_miso_t/aln2"),
              (ok_reads, good, "/tmp/_miso_t/good2")]
    batch, slots, written, *_ = sampler.prepare_batch(100, events, num_chains=1, burn_in=10, lag=1, verbose=True)
    assert [i for i, _, _, _ in slots] == [0, 3, 5, 6]
    assert len(batch) == 4
    assert [g for g, _ in sampler.skipped_genes] == ["sixty-six-isoforms", "good", "sixty-six-isoforms"]
    out = capsys.readouterr().out
    assert out.count("Skipping gene") == 3 and "isoforms" in out and "CIGAR" in out.upper()


def test_collecting_piece_by_piece_equals_collecting_at_once(tmp_path, bam_path):
    """run_miso.compute_gene_psi (round 6) collects its gene list in pieces on a thread of its own -- `entry_offset`, one
    `cache` of loaded index files, the gene objects made ahead of the alignment file (`preload_genes`): the events, their
    numbers in the caller's list (the id of their random stream) and their output names are those of one call over the whole
    list, whatever the piece size."""
    gff = tmp_path / "many.gff"
    with open(gff, "w") as g:
        g.write("##gff-version 3\n")
        for e in range(11):
            off = 10000 + 5000 * e
            ex = [(off, off + 120), (off + 400, off + 480), (off + 900, off + 1050)]
            gid = "ev%02d" % e
            g.write("10\tSE\tgene\t%d\t%d\t.\t+\t.\tID=%s;Name=%s\n" % (ex[0][0], ex[-1][1], gid, gid))
            for m, iso in enumerate(([0, 1, 2], [0, 2])):
                tid = "%s.%s" % (gid, "AB"[m])
                g.write("10\tSE\tmRNA\t%d\t%d\t.\t+\t.\tID=%s;Parent=%s\n" % (ex[iso[0]][0], ex[iso[-1]][1], tid, gid))
                for x in iso:
                    g.write("10\tSE\texon\t%d\t%d\t.\t+\t.\tID=%s.e%d;Parent=%s\n" % (ex[x][0], ex[x][1], tid, x, tid))
    idx = str(tmp_path / "indexed")
    index_gff.index_gff(str(gff), idx)
    entries = sorted(gff_utils.get_gene_ids_to_gff_index(idx).items())
    assert len(entries) == 11
    Settings.load(None)
    bam = sam_utils.Samfile(bam_path)
    out = str(tmp_path / "out")

    def describe(events):
        return [(ev[1].label, ev[2], ev[4], (ev[0].chrom, ev[0].start, ev[0].end, ev[0].min_reads), [len(i.desc) for i in ev[1].isoforms])
                for ev in events]
    whole, info = run_miso.collect_gene_events(entries, bam, out, 36, 1, verbose=False, native=True)
    assert [e[4] for e in whole] == list(range(11))
    for piece in (1, 3, 4, 11, 50):
        cache, got, infos = {}, [], {}
        run_miso.preload_genes(entries, cache)
        made = len(cache.get("genes", {}))
        for lo in range(0, len(entries), piece):
            ev, inf = run_miso.collect_gene_events(entries[lo:lo + piece], bam, out, 36, 1, verbose=False, native=True,
                                                   entry_offset=lo, cache=cache)
            got += ev
            infos.update(inf)
        assert describe(got) == describe(whole) and infos == info, piece
        assert made in (0, 11) and not cache.get("genes")      # every prepared gene object was used exactly once
    # without the preload (no bundle, or a caller that did not ask for it): the same again
    got = []
    cache = {}
    for lo in range(0, len(entries), 4):
        got += run_miso.collect_gene_events(entries[lo:lo + 4], bam, out, 36, 1, verbose=False, native=True, entry_offset=lo, cache=cache)[0]
    assert describe(got) == describe(whole)
