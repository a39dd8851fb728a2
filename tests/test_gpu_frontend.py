"""GPU: `index_gff` + `miso --run` end to end on the reference's own test data set (the inputs of
misopy/test_miso.py:131-171 test_z_gene_psi: Atp2b1 GFF + c2c12 SAM), through the CLI, one child
process per GPU -- and the result against (a) the REAL reference's run stored in the golden fixture
(posterior mean within Monte-Carlo error) and (b) the oracle's counter mode (every printed digit)."""
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
