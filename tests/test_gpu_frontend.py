"""GPU: `index_gff` + `miso --run` end to end on the reference's own test data set (the inputs of
misopy/test_miso.py:131-171 test_z_gene_psi: Atp2b1 GFF + c2c12 SAM), through the CLI, one child
process per GPU -- and the result against (a) the REAL reference's run stored in the golden fixture
(posterior mean within Monte-Carlo error) and (b) the oracle's counter mode (every printed digit)."""
import glob
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

import _golden
from _bam import sam_to_bam
from _libs import OrcLib
from _problems import flat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miso_amd"))
DATA = os.path.join(ROOT, "tests", "golden", "data")

pytestmark = pytest.mark.gpu


def run(args, **kw):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=600, **kw)


@pytest.mark.parametrize("fmt", ["bam", "sam"])
def test_miso_run_cli_on_reference_test_data(tmp_path, fmt):
    import miso_sampler
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        sam_text = f.read()
    aln = str(tmp_path / ("c2c12.Atp2b1." + fmt))
    if fmt == "bam":
        sam_to_bam(sam_text, aln)
    else:
        open(aln, "w").write(sam_text)
    idx, out = str(tmp_path / "indexed"), str(tmp_path / "out")
    settings = tmp_path / "settings.txt"
    settings.write_text("[data]\nfilter_results = True\nmin_event_reads = 20\n"
                        "[sampler]\nburn_in = 200\nlag = 4\nnum_iters = 1000\nnum_chains = 2\n")
    r = run(["-m", "miso_amd.index_gff", "--index", os.path.join(DATA, "Atp2b1.mm9.gff"), idx])
    assert r.returncode == 0, r.stdout
    r = run(["-m", "miso_amd.miso", "--run", idx, aln, "--output-dir", out, "--read-len", "36",
             "--settings-filename", str(settings), "-p", "1", "--seed", "31"])
    logs = "".join(open(os.path.join(out, "batch-logs", f)).read()
                   for f in os.listdir(os.path.join(out, "batch-logs")))
    assert r.returncode == 0, r.stdout + logs
    miso_file = os.path.join(out, "10", "ENSMUSG00000019943.miso")
    assert os.path.isfile(miso_file), r.stdout + logs
    samples, hdr, scores = miso_sampler.load_samples(miso_file)
    g = _golden.load("atp2b1")
    counts = ",".join("(%s):%d" % (",".join(str(int(v)) for v in t), c)
                      for t, c in zip(g["class_templates"], g["class_counts"]))
    assert hdr["counts"] == counts and hdr["iters"] == "1000" and hdr["burn_in"] == "200"
    assert hdr["chrom"] == "10" and hdr["strand"] == "+"
    assert hdr["mRNA_starts"] == "98377804,98377804" and hdr["mRNA_ends"] == "98457192,98486420"
    assert samples.shape == (400, 2)
    # (a) the real reference's posterior (different random stream): within Monte-Carlo error
    assert abs(samples[:, 0].mean() - g["samples"][:, 0].mean()) < 0.02
    # (b) the checker in counter mode on the same reads, gene built the way py2c_gene builds it
    # from the GFF (all exons of all transcripts; shared exons resolve to their first copy)
    from miso_amd import gene_utils
    gene = gene_utils.load_genes_from_gff(os.path.join(DATA, "Atp2b1.mm9.gff"),
                                          suppress_warnings=True)["ENSMUSG00000019943"]["gene_object"]
    exons = [(p.start, p.end) for p in gene.parts]
    isoforms = [[gene.parts.index(p) for p in iso.parts] for iso in gene.isoforms]
    orc = OrcLib()
    cpu = orc.miso(orc.gene(flat(exons), isoforms), g["pos"], g["cigars"], 36, iters=1000, burn=200,
                   lag=4, chains=2, mode=OrcLib.COUNTER, seed=31, event_id=0)
    assert cpu.rc == 0
    assert np.array_equal(samples, np.round(cpu.samples, 4))
    # a second run refuses to overwrite (miso_sampler.py:233-238) and still exits cleanly
    r = run(["-m", "miso_amd.miso", "--run", idx, aln, "--output-dir", out, "--read-len", "36",
             "--settings-filename", str(settings), "-p", "1", "--seed", "32"])
    assert r.returncode == 0
    assert np.array_equal(miso_sampler.load_samples(miso_file)[0], samples)


def test_miso_run_cli_collapsed_opt_in(tmp_path):
    """MISO_COLLAPSED=1: the same command line, the two-isoform gene through the collapsed Gibbs step
    (sampler_lane): every printed digit equals the checker's collapsed mode, the header and the read classes are the
    per-read mode's, the posterior mean agrees with the real reference's run of the golden fixture."""
    import miso_sampler
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        sam_text = f.read()
    aln = str(tmp_path / "c2c12.Atp2b1.sam")
    open(aln, "w").write(sam_text)
    idx, out = str(tmp_path / "indexed"), str(tmp_path / "out")
    settings = tmp_path / "settings.txt"
    settings.write_text("[data]\nfilter_results = True\nmin_event_reads = 20\n"
                        "[sampler]\nburn_in = 200\nlag = 4\nnum_iters = 1000\nnum_chains = 2\n")
    r = run(["-m", "miso_amd.index_gff", "--index", os.path.join(DATA, "Atp2b1.mm9.gff"), idx])
    assert r.returncode == 0, r.stdout
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MISO_COLLAPSED="1")
    r = subprocess.run([sys.executable, "-m", "miso_amd.miso", "--run", idx, aln, "--output-dir", out, "--read-len", "36",
                        "--settings-filename", str(settings), "-p", "1", "--seed", "31"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    miso_file = os.path.join(out, "10", "ENSMUSG00000019943.miso")
    assert os.path.isfile(miso_file), r.stdout
    samples, hdr, scores = miso_sampler.load_samples(miso_file)
    g = _golden.load("atp2b1")
    counts = ",".join("(%s):%d" % (",".join(str(int(v)) for v in t), c)
                      for t, c in zip(g["class_templates"], g["class_counts"]))
    assert hdr["counts"] == counts and samples.shape == (400, 2)
    assert abs(samples[:, 0].mean() - g["samples"][:, 0].mean()) < 0.02
    from miso_amd import gene_utils
    gene = gene_utils.load_genes_from_gff(os.path.join(DATA, "Atp2b1.mm9.gff"),
                                          suppress_warnings=True)["ENSMUSG00000019943"]["gene_object"]
    exons = [(p.start, p.end) for p in gene.parts]
    isoforms = [[gene.parts.index(p) for p in iso.parts] for iso in gene.isoforms]
    orc = OrcLib()
    kw = dict(iters=1000, burn=200, lag=4, chains=2, seed=31, event_id=0)
    cpu = orc.miso(orc.gene(flat(exons), isoforms), g["pos"], g["cigars"], 36, mode=OrcLib.COLLAPSED, **kw)
    per_read = orc.miso(orc.gene(flat(exons), isoforms), g["pos"], g["cigars"], 36, mode=OrcLib.COUNTER, **kw)
    assert cpu.rc == 0
    assert np.array_equal(samples, np.round(cpu.samples, 4))
    assert not np.array_equal(samples, np.round(per_read.samples, 4))   # other draws than the default mode's


def test_native_writer_and_region_path_equal_the_python_path(tmp_path):
    """One event three ways -- run_sampler (Python row loop, miso_sampler.py:456-464),
    run_sampler_batch with explicit reads and run_sampler_batch with the reads still in the
    alignment file (AlnRegion) -- must give byte-identical .miso files."""
    import miso_sampler
    from miso_amd import gene_utils, sam_utils
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        sam_text = f.read()
    aln = str(tmp_path / "reads.bam")
    sam_to_bam(sam_text, aln)
    bam = sam_utils.Samfile(aln)
    gene = gene_utils.load_genes_from_gff(os.path.join(DATA, "Atp2b1.mm9.gff"),
                                          suppress_warnings=True)["ENSMUSG00000019943"]["gene_object"]
    reads, n = bam.parse_reads("10", 98377804, 98486420, given_read_len=36)
    assert n > 3000

    def sampler():
        return miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    a = sampler().run_sampler(600, reads, gene, None, None, str(tmp_path / "a" / "g"), num_chains=3,
                              burn_in=100, lag=5, verbose=False, seed=5)
    b = sampler().run_sampler_batch(600, [(reads, gene, str(tmp_path / "b" / "g"))], num_chains=3,
                                    burn_in=100, lag=5, seed=5)[0]
    region = miso_sampler.AlnRegion(bam, "10", 98377804, 98486420, read_len=36, min_reads=20)
    c = sampler().run_sampler_batch(600, [(region, gene, str(tmp_path / "c" / "g")),
                                          (miso_sampler.AlnRegion(bam, "10", 5, 10, read_len=36, min_reads=20),
                                           gene, str(tmp_path / "c" / "empty")),
                                          (miso_sampler.AlnRegion(bam, "nope", 5, 10), gene,
                                           str(tmp_path / "c" / "nochrom"))],
                                    num_chains=3, burn_in=100, lag=5, seed=5)
    assert a and b and c[0] and c[1] is None and c[2] is None
    ref = open(a, "rb").read()
    assert open(b, "rb").read() == ref
    assert open(c[0], "rb").read() == ref
    assert ref.count(b"\n") == 2 + 3 * (600 - 100) // 5


def test_native_writer_many_isoforms_paired(tmp_path):
    """Paired-end, 5 isoforms, several events: native rows == the rows Python formats from the
    samples pysplicing returns."""
    import miso_sampler
    import pysplicing
    from miso_amd import workload
    events = []
    for e in range(3):
        exons, isoforms, pos, cig = workload.event_reads(e, K=5, n_reads=300, paired=True)
        gene = miso_sampler.SimpleGene(exons, isoforms, label="g%d" % e, chrom="1", strand="+")
        reads = (tuple(int(p) - 1 for p in pos), tuple(c.decode() for c in cig))
        events.append((reads, gene, str(tmp_path / ("e%d" % e))))
    s = miso_sampler.MISOSampler(miso_sampler.get_paired_end_sampler_params(5, 250.0, 900.0, 36),
                                 paired_end=True)
    written = s.run_sampler_batch(400, events, num_chains=2, burn_in=100, lag=3, seed=9)
    assert all(written)
    direct = pysplicing.MISOPairedBatch(
        tuple((miso_sampler.py2c_gene(g), tuple(p + 1 for p in r[0]), r[1], (1.0,) * 5)
              for r, g, _ in events), 36, 250.0, 900.0, 4.0, 400, 100, 3, 1, 2, seed=9)
    for e, res in enumerate(direct):
        psi = np.transpose(np.array(res[0]))
        rows = ["%s\t%.2f" % (",".join("%.4f" % v for v in p), ll) for p, ll in zip(psi, res[1])]
        got = open(written[e]).read().split("\n")
        assert got[2:-1] == rows


def test_event_ids_pin_the_random_stream(tmp_path):
    """An event numbered g gives the same file whether it runs alone (first_event_id = g), inside a
    batch next to others, or after events that the skip rules dropped."""
    import miso_sampler
    from miso_amd import workload
    evs = []
    for e in range(3):
        exons, isoforms, pos, cig = workload.event_reads(e, K=2 + e, n_reads=200)
        gene = miso_sampler.SimpleGene(exons, isoforms, label="g%d" % e)
        evs.append(((tuple(int(p) - 1 for p in pos), tuple(c.decode() for c in cig)), gene))

    def sampler():
        return miso_sampler.MISOSampler(miso_sampler.get_single_end_sampler_params(2, 36))
    kw = dict(num_chains=2, burn_in=50, lag=2, seed=4)
    alone = [sampler().run_sampler_batch(250, [(r, g, str(tmp_path / "a" / ("e%d" % i)))], first_event_id=10 * i + 7, **kw)[0]
             for i, (r, g) in enumerate(evs)]
    empty = (((), ()), evs[0][1], str(tmp_path / "b" / "skipped"), None, 3)
    together = sampler().run_sampler_batch(
        250, [empty] + [(r, g, str(tmp_path / "b" / ("e%d" % i)), None, 10 * i + 7) for i, (r, g) in enumerate(evs)],
        first_event_id=1000, **kw)
    assert together[0] is None
    for a, b in zip(alone, together[1:]):
        assert open(a, "rb").read() == open(b, "rb").read()


def test_cli_summarize_and_compare(tmp_path):
    """`miso --run ... --summarize` and `miso --run ... --compare BAM2`: the summarize_miso and
    compare_miso tables (samples_utils.py:263-329, hypothesis_test.py:186-345) straight from the
    run, numbers computed on the GPU."""
    import miso_sampler
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        lines = f.read().splitlines()
    head = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if not l.startswith("@")]
    bam1, bam2 = str(tmp_path / "s1.bam"), str(tmp_path / "s2.bam")
    sam_to_bam("\n".join(head + body) + "\n", bam1)
    # sample 2: drop most reads that only fit the long isoform -> psi shifts
    keep = [l for i, l in enumerate(body) if "N" in l.split("\t")[5] or i % 3 == 0]
    sam_to_bam("\n".join(head + keep) + "\n", bam2)
    idx = str(tmp_path / "indexed")
    settings = tmp_path / "settings.txt"
    settings.write_text("[data]\nmin_event_reads = 20\n[sampler]\nburn_in = 200\nlag = 4\nnum_iters = 1000\nnum_chains = 2\n")
    assert run(["-m", "miso_amd.index_gff", "--index", os.path.join(DATA, "Atp2b1.mm9.gff"), idx]).returncode == 0
    out = str(tmp_path / "control")
    r = run(["-m", "miso_amd.miso", "--run", idx, bam1, "--output-dir", out, "--read-len", "36",
             "--settings-filename", str(settings), "-p", "1", "--seed", "31", "--summarize"])
    assert r.returncode == 0, r.stdout
    table = os.path.join(out, "summary", "control.miso_summary")
    rows = [l.rstrip("\n").split("\t") for l in open(table)]
    assert rows[0][:4] == ["event_name", "miso_posterior_mean", "ci_low", "ci_high"] and len(rows) == 2
    assert rows[1][0] == "ENSMUSG00000019943"
    samples, hdr, _ = miso_sampler.load_samples(os.path.join(out, "10", "ENSMUSG00000019943.miso"))
    assert abs(float(rows[1][1]) - samples[:, 0].mean()) < 0.006
    assert float(rows[1][2]) <= float(rows[1][1]) <= float(rows[1][3]) and rows[1][5] == hdr["counts"]
    out2 = str(tmp_path / "cmp")
    r = run(["-m", "miso_amd.miso", "--run", idx, bam1, "--compare", bam2, "--labels", "ctl", "kd",
             "--output-dir", out2, "--read-len", "36", "--settings-filename", str(settings), "-p", "1",
             "--seed", "31"])
    logs = "".join(open(os.path.join(out2, "batch-logs", f)).read() for f in os.listdir(os.path.join(out2, "batch-logs")))
    assert r.returncode == 0, r.stdout + logs
    bf = os.path.join(out2, "ctl_vs_kd", "bayes-factors", "ctl_vs_kd.miso_bf")
    rows = [l.rstrip("\n").split("\t") for l in open(bf)]
    assert rows[0][0] == "event_name" and rows[0][8] == "bayes_factor" and len(rows) == 2, logs
    for lab in ("ctl", "kd"):
        assert os.path.isfile(os.path.join(out2, lab, "10", "ENSMUSG00000019943.miso"))
    m1, m2, diff, bfv = float(rows[1][1]), float(rows[1][4]), float(rows[1][7]), float(rows[1][8])
    assert abs((m1 - m2) - diff) < 0.011 and bfv >= 0
    s1 = miso_sampler.load_samples(os.path.join(out2, "ctl", "10", "ENSMUSG00000019943.miso"))[0]
    assert abs(m1 - s1[:, 0].mean()) < 0.006


def test_summarize_and_compare_existing_miso_directories(tmp_path):
    """summarize_miso --summarize-samples / compare_miso --compare-samples on directories of `.miso` files
    (misopy/samples_utils.py:263-329, hypothesis_test.py:186-345): the table written from the files equals, line
    for line, the table the live run wrote (which summarises the same text, miso_batch_summarize_as_text); the
    comparison's means / bounds equal the numpy restatement on the parsed files, its Bayes factors the
    scipy-pinned checker within the float tolerance of tests/test_gpu_compare.py."""
    import miso_sampler
    from _compare_ref import bayes_factor
    from _summary_ref import credible_interval
    from miso_amd import samples_utils
    from miso_sampler import SimpleGene
    rng = np.random.default_rng(5)
    dirs = []
    for label, shift in (("ctl", 0.0), ("kd", 0.25)):
        out = tmp_path / label
        params = miso_sampler.get_single_end_sampler_params(2, 36, 1)
        sampler = miso_sampler.MISOSampler(params, paired_end=False)
        events = []
        for e in range(70):                       # > 64 files: the parallel parser
            K = 2 + (e % 3)
            exons = [(1 + 200 * i, 100 + 200 * i) for i in range(K + 1)]
            isoforms = [list(range(K + 1))] + [[x for x in range(K + 1) if x != k] for k in range(1, K)]
            gene = SimpleGene(exons, isoforms, label="ev%03d" % e, chrom="chr%d" % (e % 3))
            n = 60 + 7 * e
            pos = rng.integers(1, 200 * K + 60, size=n)
            if shift:
                pos = np.where(rng.random(n) < shift, rng.integers(1, 60, size=n), pos)
            events.append(((list(int(x) for x in pos), ["36M"] * n), gene, str(out / gene.chrom / gene.label)))
        sampler.run_sampler_batch(600, events, num_chains=2, burn_in=100, lag=2, seed=9, first_event_id=0,
                                  summary_file=str(tmp_path / (label + ".live_summary")))
        dirs.append(str(out))
    for label, d in zip(("ctl", "kd"), dirs):
        table = str(tmp_path / "sum" / "summary" / (label + ".miso_summary"))
        assert samples_utils.main(["--summarize-samples", d, str(tmp_path / "sum")]) == 0
        live = sorted(open(str(tmp_path / (label + ".live_summary"))).read().splitlines()[1:])
        walked = sorted(open(table).read().splitlines()[1:])
        assert walked == live and len(walked) > 30
    assert samples_utils.main(["--compare-samples", dirs[0], dirs[1], str(tmp_path / "cmp")]) == 0
    rows = [l.split("\t") for l in open(str(tmp_path / "cmp" / "ctl_vs_kd" / "bayes-factors" / "ctl_vs_kd.miso_bf")).read().splitlines()]
    assert rows[0][0] == "event_name" and rows[0][8] == "bayes_factor" and len(rows) > 30
    checked = 0
    for r in rows[1:]:
        s1 = samples_utils.parse_miso_file(glob.glob(os.path.join(dirs[0], "*", r[0] + ".miso"))[0])[1]
        s2 = samples_utils.parse_miso_file(glob.glob(os.path.join(dirs[1], "*", r[0] + ".miso"))[0])[1]
        K = s1.shape[1]
        lo1 = [credible_interval(s1[:, k])[0] for k in range(K)]
        assert r[2] == (",".join("%.2f" % v for v in lo1) if K > 2 else "%.2f" % lo1[0])
        bfs = [bayes_factor(s1[:, k], s2[:, k], 0.3)[0] for k in range(K)]
        want = ",".join("%.2f" % max(v, 0) for v in bfs) if K > 2 else "%.2f" % bfs[0]
        got = [float(x) for x in r[8].split(",")]
        for g, w in zip(got, [float(x) for x in want.split(",")]):
            assert abs(g - w) <= 0.011 + 1e-6 * abs(w), (r[0], r[8], want)
        checked += 1
    assert checked > 30


def test_results_do_not_depend_on_the_number_of_worker_processes(tmp_path):
    """`miso --run -p 1` and `-p 3` (three chunks, here sharing the one GPU) over 7 genes, two of which
    the skip rules drop: every .miso file identical -- each gene keeps its global number in the
    random-number counter whatever the split."""
    from miso_amd import workload
    gff, sam = tmp_path / "g.gff", tmp_path / "r.sam"
    lines, recs = ["##gff-version 3"], []
    for e in range(7):
        off = 10000 + e * 6000
        exons, isoforms, pos, cig = workload.event_reads(e, 2 + (e % 3), 15 if e in (2, 5) else 300)
        ex = [(s + off, t + off) for s, t in exons]
        gid = "gene%d" % e
        lines.append("chr1\tx\tgene\t%d\t%d\t.\t+\t.\tID=%s" % (ex[0][0], ex[-1][1], gid))
        for m, iso in enumerate(isoforms):
            tid = "%s.t%d" % (gid, m)
            lines.append("chr1\tx\tmRNA\t%d\t%d\t.\t+\t.\tID=%s;Parent=%s" % (ex[iso[0]][0], ex[iso[-1]][1], tid, gid))
            lines += ["chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=%s.e%d;Parent=%s" % (ex[x][0], ex[x][1], tid, x, tid) for x in iso]
        recs += ["r%d_%d\t0\tchr1\t%d\t255\t%s\t*\t0\t0\t%s\t%s" % (e, i, pos[i] + off, cig[i].decode(), "A" * 36, "I" * 36)
                 for i in range(len(pos))]
    gff.write_text("\n".join(lines) + "\n")
    sam.write_text("@SQ\tSN:chr1\tLN:100000\n" + "\n".join(recs) + "\n")
    settings = tmp_path / "s.txt"
    settings.write_text("[data]\nmin_event_reads = 20\n[sampler]\nburn_in = 100\nlag = 5\nnum_iters = 600\nnum_chains = 2\n")
    idx = str(tmp_path / "idx")
    assert run(["-m", "miso_amd.index_gff", "--index", str(gff), idx]).returncode == 0
    outs = {}
    for nproc in (1, 3):
        out = str(tmp_path / ("out%d" % nproc))
        r = run(["-m", "miso_amd.miso", "--run", idx, str(sam), "--output-dir", out, "--read-len", "36",
                 "--settings-filename", str(settings), "-p", str(nproc), "--seed", "77"])
        assert r.returncode == 0, r.stdout
        outs[nproc] = {f: open(os.path.join(out, "chr1", f), "rb").read() for f in sorted(os.listdir(os.path.join(out, "chr1")))}
    assert sorted(outs[1]) == ["gene%d.miso" % e for e in (0, 1, 3, 4, 6)]      # 15-read genes skipped
    assert outs[1] == outs[3]


def _seven_genes(tmp_path, tag, small=(2, 5), n_reads=300, shift=0):
    """GFF3 + SAM of 7 synthetic genes (2-4 isoforms); the genes in `small` get 15 reads (below
    min_event_reads = 20: the skip rules drop them)."""
    from miso_amd import workload
    gff, sam = tmp_path / ("g%s.gff" % tag), tmp_path / ("r%s.sam" % tag)
    lines, recs = ["##gff-version 3"], []
    for e in range(7):
        off = 10000 + e * 6000
        exons, isoforms, pos, cig = workload.event_reads(e + shift, 2 + (e % 3), 15 if e in small else n_reads)
        ex = [(s + off, t + off) for s, t in workload.event_gene(e, 2 + (e % 3))[0]]
        gid = "gene%d" % e
        lines.append("chr1\tx\tgene\t%d\t%d\t.\t+\t.\tID=%s" % (ex[0][0], ex[-1][1], gid))
        for m, iso in enumerate(isoforms):
            tid = "%s.t%d" % (gid, m)
            lines.append("chr1\tx\tmRNA\t%d\t%d\t.\t+\t.\tID=%s;Parent=%s" % (ex[iso[0]][0], ex[iso[-1]][1], tid, gid))
            lines += ["chr1\tx\texon\t%d\t%d\t.\t+\t.\tID=%s.e%d;Parent=%s" % (ex[x][0], ex[x][1], tid, x, tid) for x in iso]
        recs += ["r%d_%d\t0\tchr1\t%d\t255\t%s\t*\t0\t0\t%s\t%s" % (e, i, pos[i] + off, cig[i].decode(), "A" * 36, "I" * 36)
                 for i in range(len(pos))]
    gff.write_text("\n".join(lines) + "\n")
    sam.write_text("@SQ\tSN:chr1\tLN:100000\n" + "\n".join(recs) + "\n")
    return gff, sam


def test_compare_results_do_not_depend_on_the_number_of_worker_processes(tmp_path):
    """`miso --run ... --compare BAM2` with -p 1 and -p 3 over 7 genes, one skipped in sample 1 and another
    in sample 2: both samples' .miso files and the .miso_bf table are byte-identical whatever the split,
    and sample 1's files equal those of a plain `miso --run` of the same alignments -- every gene keeps
    its number in the full gene list as its id in the random-number counter (the compare
    path numbered the genes by their position among the kept pairs)."""
    gff, sam1 = _seven_genes(tmp_path, "1", small=(2,))
    _, sam2 = _seven_genes(tmp_path, "2", small=(5,), n_reads=280)
    settings = tmp_path / "s.txt"
    settings.write_text("[data]\nmin_event_reads = 20\n[sampler]\nburn_in = 100\nlag = 5\nnum_iters = 600\nnum_chains = 2\n")
    idx = str(tmp_path / "idx")
    assert run(["-m", "miso_amd.index_gff", "--index", str(gff), idx]).returncode == 0
    outs = {}
    for nproc in (1, 3):
        out = str(tmp_path / ("cmp%d" % nproc))
        r = run(["-m", "miso_amd.miso", "--run", idx, str(sam1), "--compare", str(sam2), "--labels", "a", "b",
                 "--output-dir", out, "--read-len", "36", "--settings-filename", str(settings), "-p", str(nproc),
                 "--seed", "77"])
        assert r.returncode == 0, r.stdout
        files = {}
        for lab in ("a", "b"):
            d = os.path.join(out, lab, "chr1")
            files.update({lab + "/" + f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))})
        files["bf"] = open(os.path.join(out, "a_vs_b", "bayes-factors", "a_vs_b.miso_bf"), "rb").read()
        outs[nproc] = files
    kept = [e for e in range(7) if e not in (2, 5)]
    assert sorted(k for k in outs[1] if k.startswith("a/")) == ["a/gene%d.miso" % e for e in kept]
    assert outs[1] == outs[3]
    plain = str(tmp_path / "plain")
    r = run(["-m", "miso_amd.miso", "--run", idx, str(sam1), "--output-dir", plain, "--read-len", "36",
             "--settings-filename", str(settings), "-p", "2", "--seed", "77"])
    assert r.returncode == 0, r.stdout
    for e in kept:
        assert open(os.path.join(plain, "chr1", "gene%d.miso" % e), "rb").read() == outs[1]["a/gene%d.miso" % e], e


def test_cli_summary_only_writes_the_same_table_and_no_miso_files(tmp_path):
    """`miso --run ... --summary-only` (round 6): the `summarize_miso` table of the run without the per-event `.miso`
    files -- byte for byte the table `--summarize` writes beside its files (both summarise the four-decimal text a file
    holds, samples_utils.py:263-329, on the device; same seed, same event numbers), through the four-stage pipeline of
    run_miso.compute_gene_psi with the worker opening the alignment file itself."""
    with gzip.open(os.path.join(DATA, "c2c12.Atp2b1.sam.gz"), "rt") as f:
        sam_text = f.read()
    bam = str(tmp_path / "reads.bam")
    sam_to_bam(sam_text, bam)
    idx = str(tmp_path / "indexed")
    settings = tmp_path / "settings.txt"
    settings.write_text("[data]\nmin_event_reads = 20\n[sampler]\nburn_in = 200\nlag = 4\nnum_iters = 1000\nnum_chains = 2\n")
    assert run(["-m", "miso_amd.index_gff", "--index", os.path.join(DATA, "Atp2b1.mm9.gff"), idx]).returncode == 0
    tables = {}
    for mode in ("--summarize", "--summary-only"):
        out = str(tmp_path / ("out" + mode.strip("-")))
        r = run(["-m", "miso_amd.miso", "--run", idx, bam, "--output-dir", out, "--read-len", "36",
                 "--settings-filename", str(settings), "-p", "1", "--seed", "31", mode])
        logs = "".join(open(os.path.join(out, "batch-logs", f)).read() for f in os.listdir(os.path.join(out, "batch-logs")))
        assert r.returncode == 0, r.stdout + logs
        table = glob.glob(os.path.join(out, "summary", "*.miso_summary"))
        assert len(table) == 1, logs
        tables[mode] = open(table[0]).read()
        files = glob.glob(os.path.join(out, "**", "*.miso"), recursive=True)
        assert (len(files) == 1) == (mode == "--summarize"), (mode, files)
    assert tables["--summarize"] == tables["--summary-only"]
    assert tables["--summary-only"].count("\n") == 2 and "ENSMUSG00000019943" in tables["--summary-only"]
