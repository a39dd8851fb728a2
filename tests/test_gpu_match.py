"""Row f1: read x isoform compatibility computed on the GPU (kernels_match.hip) against the golden
vectors made from the real reference (solve.c:8-108, 141-218 outputs in tests/golden/*.npz), the
CPU checker and the library's own host path.  Integer work: bit-exact."""
import numpy as np
import pytest

import _golden
from _problems import se_gene, expr_for, flat
from miso_amd import capi

pytestmark = pytest.mark.gpu


def _dev_batch(g, paired=False, overhang=1, **kw):
    if paired:
        kw.update(mean=float(g["mean"]), var=float(g["var"]))
    return capi.Batch(g["read_len"], iters=g["iters"], burn=g["burn"], lag=g["lag"], chains=g["chains"],
                      overhang=overhang, paired=paired, counts_trace=True, device_match=True, **kw)


@pytest.mark.parametrize("name", _golden.names("se"))
def test_device_match_equals_reference_single_end(name):
    g = _golden.load(name)
    if "match" not in g and "match_ov1" not in g:
        pytest.skip("fixture without a match matrix")
    G = capi.Gene(g["exon_list"], g["isoform_list"])
    for ov in ((1, 4) if "match_ov1" in g else (g["overhang"],)):
        b = _dev_batch(g, overhang=ov)
        i = b.add_event(G, g["pos"], g["cigars"])
        b.upload(0)
        m, _ = b.device_match_of(i)
        want = g["match_ov%d" % ov] if "match_ov1" in g else g["match"]
        assert np.array_equal(m, want)
        assert np.array_equal(m, G.match_iso(g["pos"], g["cigars"], g["read_len"], overhang=ov))
        if "class_templates" in g and ov == g["overhang"]:
            ct, cc = b.classes(i)
            assert np.array_equal(ct, g["class_templates"]) and np.array_equal(cc, g["class_counts"])


@pytest.mark.parametrize("name", _golden.names("pe"))
def test_device_match_equals_reference_paired_end(name):
    g = _golden.load(name)
    G = capi.Gene(g["exon_list"], g["isoform_list"])
    b = _dev_batch(g, paired=True, overhang=g["overhang"])
    i = b.add_event(G, g["pos"], g["cigars"])
    b.upload(0)
    m, fl = b.device_match_of(i)
    assert np.array_equal(m, g["match"]) and np.array_equal(fl, g["fraglen"])
    ct, cc = b.classes(i)
    assert np.array_equal(ct, g["class_templates"]) and np.array_equal(cc, g["class_counts"])


@pytest.mark.parametrize("paired", [False, True])
def test_device_match_batch_equals_host_path(paired, orc):
    """Many events of mixed K in one launch: same problems, same sampler output as host matching."""
    kw = dict(iters=300, burn=100, lag=2, chains=2, paired=paired)
    if paired:
        kw.update(mean=250.0, var=900.0)
    dev = capi.Batch(36, counts_trace=True, device_match=True, **kw)
    host = capi.Batch(36, counts_trace=True, **kw)
    specs = [(K, 40 + 37 * j, 100 * K + j) for K in (2, 3, 5, 10, 2, 7) for j in range(3)] + [(4, 0, 1), (3, 1, 2)]
    keep = []
    for K, n, sd in specs:
        exons, isoforms = se_gene(K, exlen=500, gap=300) if paired else se_gene(K)
        G = capi.Gene(exons, isoforms)
        keep.append(G)
        _, pos, cig = capi.simulate_reads(G, expr_for(K), n, 36, sd, 250.0 if paired else 0.0, 900.0 if paired else 0.0)
        assert dev.add_event(G, pos, cig) == host.add_event(G, pos, cig)
        og = orc.gene(flat(exons), isoforms)
        if not paired and n:
            assert np.array_equal(G.match_iso(pos, cig, 36), orc.match_iso(og, pos, cig, 36, 1)[1])
    dev.run(seed=5); host.run(seed=5)
    for i, (K, n, sd) in enumerate(specs):
        m, fl = dev.device_match_of(i)
        G = keep[i]
        a, b = dev.result(i, trace=True), host.result(i, trace=True)
        assert np.array_equal(a.samples, b.samples) and np.array_equal(a.loglik, b.loglik)
        assert np.array_equal(a.assignment, b.assignment) and np.array_equal(a.counts_hash, b.counts_hash)
        assert np.array_equal(a.class_templates, b.class_templates) and np.array_equal(a.class_counts, b.class_counts)
    assert dev.match_ms() > 0


def test_device_match_errors_surface_at_add_event():
    b = capi.Batch(36, iters=100, burn=10, lag=1, chains=1, device_match=True)
    G = capi.Gene([(1, 100), (201, 300)], [[0, 1], [0]])
    with pytest.raises(capi.InternalError, match="Unsupported CIGAR"):
        b.add_event(G, [10], [b"36Q"])
    with pytest.raises(capi.InternalError, match="Bad CIGAR string"):
        b.add_event(G, [10], [b"10M5S10M"])
    assert len(b) == 0
    b.add_event(G, [10, 50], [b"36M", b"36M"])
    assert b.classes(0)[1].size == 0          # read classes exist only after the upload
    b.run(seed=1)
    assert b.classes(0)[1].sum() == 2
