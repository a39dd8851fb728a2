#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the authoring container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Each fixture is an .npz holding the inputs (gene, read positions, CIGARs, sampler parameters,
MT19937 seed) and what the reference C core (oracle/_ref/libmiso_ref.so = the reference's own
sources behind oracle/ref_shim.c) returned for them: match matrix, read classes, psi samples, log
scores, final assignment, run data.  tests/test_oracle_golden.py replays the inputs through
oracle/miso_oracle.c (stream mode) and demands identical bits, with or without the reference
library being present.  Data only: no reference source text is stored.

Also extracts the reference's own test data set (misopy/test-data/sam-data/c2c12.Atp2b1.sam +
misopy/gff-events/mm9/genes/Atp2b1.mm9.gff, the inputs of misopy/test_miso.py:131-171) into
plain arrays: read start (SAM POS, 1-based), CIGAR string, and the two mRNAs' exon lists.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import RefLib  # noqa: E402
from _problems import expr_for, flat, se_gene  # noqa: E402

REF_ROOT = "/root/reference/misopy"


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes")


def pack_result(r):
    return dict(match=r.match, class_templates=r.class_templates, class_counts=r.class_counts,
                samples=r.samples, loglik=r.loglik, assignment=r.assignment, rundata=r.rundata)


def iso_array(isoforms):
    out = []
    for iso in isoforms:
        out.extend(iso)
        out.append(-1)
    return np.asarray(out, np.int32)


def se_case(R, name, K, N, seed, iters, burn, lag, chains, overhang=1, read_len=36, **gene_kw):
    exons, isoforms = se_gene(K, **gene_kw)
    g = R.gene(flat(exons), isoforms)
    R.rng_seed(seed)
    stream = np.array([R.unif01() for _ in range(8)] + [R.normal01() for _ in range(8)])
    R.rng_seed(seed)
    rc, iso, pos, cig = R.simulate_reads(g, expr_for(K), N, read_len)
    assert rc == 0
    r = R.miso(g, pos, cig, read_len, iters=iters, burn=burn, lag=lag, chains=chains,
               overhang=overhang)
    assert r.rc == 0
    save(name, kind="se", exons=np.asarray(exons, np.int32), isoforms=iso_array(isoforms),
         expr=expr_for(K), seed=seed, read_len=read_len, overhang=overhang, iters=iters, burn=burn,
         lag=lag, chains=chains, pos=pos, cigars=np.array(cig), sim_isoform=iso, rng_stream=stream,
         **pack_result(r))


def pe_case(R, name, K, N, seed, iters, burn, lag, chains, mean=250.0, var=900.0, read_len=36):
    exons, isoforms = se_gene(K, exlen=500, gap=300)
    g = R.gene(flat(exons), isoforms)
    R.rng_seed(seed)
    rc, iso, pos, cig = R.simulate_paired_reads(g, expr_for(K), N, read_len, mean, var)
    assert rc == 0
    rcm, m, fl = R.match_iso_paired(g, pos, cig, read_len, mean, var)
    r = R.miso_paired(g, pos, cig, read_len, mean, var, iters=iters, burn=burn, lag=lag,
                      chains=chains)
    assert r.rc == 0 and rcm == 0
    save(name, kind="pe", exons=np.asarray(exons, np.int32), isoforms=iso_array(isoforms),
         expr=expr_for(K), seed=seed, read_len=read_len, overhang=1, iters=iters, burn=burn,
         lag=lag, chains=chains, mean=mean, var=var, pos=pos, cigars=np.array(cig),
         sim_isoform=iso, fraglen=fl, **pack_result(r))


def convergent_case(R, name, paired, K, N, seed, iters, burn, lag, chains, max_iters, mean=250.0, var=900.0,
                    read_len=36):
    """stop=CONVERGENT_MEAN (miso.c:903-925 / miso_paired.c:501-523): schedules short enough that the first rounds do
    not converge, so the fixture holds the LAST noSamples of a later round (miso.c:976-983)."""
    exons, isoforms = se_gene(K, exlen=500, gap=300) if paired else se_gene(K)
    g = R.gene(flat(exons), isoforms)
    R.rng_seed(seed)
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains)
    if paired:
        rc, iso, pos, cig = R.simulate_paired_reads(g, expr_for(K), N, read_len, mean, var)
        r = R.miso_paired(g, pos, cig, read_len, mean, var, stop=1, max_iters=max_iters, **kw)
        extra = dict(mean=mean, var=var)
    else:
        rc, iso, pos, cig = R.simulate_reads(g, expr_for(K), N, read_len)
        r = R.miso(g, pos, cig, read_len, stop=1, max_iters=max_iters, **kw)
        extra = {}
    assert rc == 0 and r.rc == 0
    save(name, kind="pe_conv" if paired else "se_conv", exons=np.asarray(exons, np.int32),
         isoforms=iso_array(isoforms), expr=expr_for(K), seed=seed, read_len=read_len, overhang=1, stop=1,
         max_iters=max_iters, pos=pos, cigars=np.array(cig), **kw, **extra, **pack_result(r))


def marginal_case(R, name, K, N, seed, iters, burn, lag, chains, overhang=1, read_len=36, stop=0, max_iters=100000):
    """algorithm=MARGINAL (miso.c:272-283, 800-808): no assignments, the match matrix divided by the effective lengths
    (which is also how the reference hands it back), one reassignment at the end for the returned vector."""
    exons, isoforms = se_gene(K)
    g = R.gene(flat(exons), isoforms)
    R.rng_seed(seed)
    rc, iso, pos, cig = R.simulate_reads(g, expr_for(K), N, read_len)
    r = R.miso(g, pos, cig, read_len, iters=iters, burn=burn, lag=lag, chains=chains, overhang=overhang, algo=1,
               stop=stop, max_iters=max_iters)
    assert rc == 0 and r.rc == 0
    save(name, kind="se_marginal", exons=np.asarray(exons, np.int32), isoforms=iso_array(isoforms), expr=expr_for(K),
         seed=seed, read_len=read_len, overhang=overhang, iters=iters, burn=burn, lag=lag, chains=chains, stop=stop,
         max_iters=max_iters, pos=pos, cigars=np.array(cig), **pack_result(r))


def marginal_cases(R):
    marginal_case(R, "se_k2_marginal", 2, 300, 53, 600, 100, 2, 2)
    marginal_case(R, "se_k3_marginal", 3, 300, 59, 500, 100, 5, 3, overhang=4)
    marginal_case(R, "se_k6_marginal", 6, 500, 61, 400, 100, 3, 2)
    marginal_case(R, "se_k3_marginal_convergent", 3, 200, 67, 50, 10, 1, 4, stop=1, max_iters=700)


def assignment_matrix_case(R):
    """The reference's splicing_assignment_matrix (assignment.c:90-276) for the skipped-exon family and a few irregular
    structures: what algorithm=CLASSES sums over.  (Its sampler on that path reads uninitialised counts, miso.c:790, so
    no CLASSES run is stored; the matrix and the score are the pins.)"""
    genes = [se_gene(K) for K in (2, 3, 5, 8, 12)]
    genes.append(([(1, 81), (115, 201), (326, 495), (631, 684), (786, 921)], [[1, 3], [1, 2, 4], [0, 1, 2, 3], [0, 4]]))
    genes.append(([(1, 60), (100, 180), (400, 520), (700, 760)], [[0, 1, 2, 3], [0, 3], [1, 2], [0, 2, 3], [2, 3]]))
    out = {}
    for i, (exons, isoforms) in enumerate(genes):
        g = R.gene(flat(exons), isoforms)
        out["exons_%d" % i] = np.asarray(exons, np.int32)
        out["isoforms_%d" % i] = iso_array(isoforms)
        out["matrix_%d" % i] = R.assignment_matrix(g, 36)
    save("assignment_matrix", kind="assignment", n=len(genes), **out)


# The five two-isoform alternative-splicing classes of BASELINE configs[2]: skipped exon, retained intron, alternative 3'
# and 5' splice sites, mutually exclusive exons -- exons that overlap or nest inside one gene (solve.c:8-108, 141-218,
# gff.c:1041-1084), which the skipped-exon family above never has.  `ri_first` lists the spanning exon first.
AS_GENES = {
    "ri": ([(1, 150), (351, 500), (1, 500)], [[0, 1], [2]]),
    "ri_first": ([(1, 500), (1, 150), (351, 500)], [[1, 2], [0]]),
    "a3ss": ([(1, 150), (401, 600), (451, 600)], [[0, 1], [0, 2]]),
    "a5ss": ([(1, 200), (1, 150), (401, 600)], [[0, 2], [1, 2]]),
    "mxe": ([(1, 100), (201, 300), (401, 500), (601, 700)], [[0, 1, 3], [0, 2, 3]]),
}


def as_case(R, name, paired, seed, N=300, iters=600, burn=100, lag=2, chains=2, mean=120.0, var=400.0, read_len=36,
            overhang=1):
    exons, isoforms = AS_GENES[name]
    g = R.gene(flat(exons), isoforms)
    R.rng_seed(seed)
    expr = np.array([0.35, 0.65])
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains, overhang=overhang)
    if paired:
        rc, iso, pos, cig = R.simulate_paired_reads(g, expr, N, read_len, mean, var)
        assert rc == 0
        rcm, m, fl = R.match_iso_paired(g, pos, cig, read_len, mean, var, overhang=overhang)
        r = R.miso_paired(g, pos, cig, read_len, mean, var, **kw)
        assert r.rc == 0 and rcm == 0
        extra = dict(mean=mean, var=var, fraglen=fl)
    else:
        rc, iso, pos, cig = R.simulate_reads(g, expr, N, read_len)
        assert rc == 0
        r = R.miso(g, pos, cig, read_len, **kw)
        assert r.rc == 0
        extra = {}
    save("as_%s_%s" % (name, "pe" if paired else "se"), kind="pe" if paired else "se",
         exons=np.asarray(exons, np.int32), isoforms=iso_array(isoforms), expr=expr, seed=seed, read_len=read_len,
         pos=pos, cigars=np.array(cig), sim_isoform=iso, **kw, **extra, **pack_result(r))


def as_cases(R):
    for j, name in enumerate(AS_GENES):
        as_case(R, name, False, 101 + j, overhang=1 if j % 2 == 0 else 4)
        as_case(R, name, True, 151 + j)


def cigar_edge_case(R):
    """Hand-written alignments exercising solve.c:220-306 / 8-108: clips, =, X, D, I, skips that
    do and do not match the annotation, overhang violations, short reads, reads off the gene."""
    exons = [(101, 200), (301, 400), (501, 600)]
    isoforms = [[0, 1, 2], [0, 2]]
    g = R.gene(flat(exons), isoforms)
    reads = [
        (120, "36M"), (180, "21M100N15M"), (180, "21M300N15M"), (190, "11M100N25M"),
        (199, "2M100N34M"), (350, "36M"), (380, "21M100N15M"), (120, "4S32M"), (120, "32M4S"),
        (120, "2H34M"), (120, "10M2I26M"), (120, "10M2D24M"), (120, "30=6X"), (120, "20M"),
        (50, "36M"), (590, "36M"), (180, "21M99N15M"), (181, "20M100N16M"), (165, "36M"),
        (166, "35M100N1M"), (120, "36M10M"), (400, "1M100N35M"), (365, "36M"),
    ]
    pos = np.array([p for p, _ in reads], np.int32)
    cig = [c.encode() for _, c in reads]
    out = {}
    for ov in (1, 4):
        rc, m = R.match_iso(g, pos, cig, 36, overhang=ov)
        assert rc == 0
        out["match_ov%d" % ov] = m
    R.rng_seed(9)
    r = R.miso(g, pos, cig, 36, iters=300, burn=50, lag=2, chains=2, overhang=1)
    assert r.rc == 0
    save("cigar_edges", kind="se", exons=np.asarray(exons, np.int32), isoforms=iso_array(isoforms),
         seed=9, read_len=36, overhang=1, iters=300, burn=50, lag=2, chains=2, pos=pos,
         cigars=np.array(cig), **out, **pack_result(r))


def atp2b1_case(R):
    """The reference's own test inputs (misopy/test_miso.py:131-171), as arrays."""
    gff = os.path.join(REF_ROOT, "gff-events/mm9/genes/Atp2b1.mm9.gff")
    sam = os.path.join(REF_ROOT, "test-data/sam-data/c2c12.Atp2b1.sam")
    mrnas, order = {}, []
    for line in open(gff):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        attrs = dict(kv.split("=", 1) for kv in f[8].split(";") if "=" in kv)
        if f[2] == "mRNA":
            mrnas[attrs["ID"]] = []
            order.append(attrs["ID"])
        elif f[2] == "exon":
            mrnas[attrs["Parent"]].append((int(f[3]), int(f[4])))
    parts = sorted({e for m in order for e in mrnas[m]})       # misopy/Gene.py: parts by start
    isoforms = [sorted(parts.index(e) for e in mrnas[m]) for m in order]
    pos, cig = [], []
    for line in open(sam):
        if line.startswith("@"):
            continue
        f = line.split("\t")
        if f[5] == "*":
            continue
        pos.append(int(f[3]))
        cig.append(f[5].encode())
    pos = np.asarray(pos, np.int32)
    g = R.gene(flat(parts), isoforms)
    R.rng_seed(42)
    r = R.miso(g, pos, cig, 36, iters=1000, burn=200, lag=4, chains=2, overhang=1)
    assert r.rc == 0
    save("atp2b1", kind="se", exons=np.asarray(parts, np.int32), isoforms=iso_array(isoforms),
         mrna_ids=np.array(order), seed=42, read_len=36, overhang=1, iters=1000, burn=200, lag=4,
         chains=2, pos=pos, cigars=np.array(cig), **pack_result(r))


def main():
    if not RefLib.available():
        sys.exit("oracle/_ref/libmiso_ref.so missing: run `make -C oracle ref` first")
    R = RefLib()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the reference prints "no chains: %d" (miso.c:837)
    try:
        if sys.argv[1:] == ["assignment"]:   # likewise
            assignment_matrix_case(R)
            return
        if sys.argv[1:] == ["marginal"]:     # likewise
            marginal_cases(R)
            return
        if sys.argv[1:] == ["as"]:           # round 5: the AS-class geometries (the others are unchanged)
            as_cases(R)
            return
        if sys.argv[1:] == ["convergent"]:   # only the fixtures added in round 4 (the others are unchanged)
            convergent_case(R, "se_k3_convergent", False, 3, 200, 37, 50, 10, 1, 4, 700)
            convergent_case(R, "se_k2_convergent", False, 2, 300, 41, 60, 20, 2, 3, 2000)
            convergent_case(R, "pe_k2_convergent", True, 2, 200, 43, 60, 20, 2, 3, 1500)
            convergent_case(R, "pe_k4_convergent", True, 4, 200, 47, 80, 30, 2, 2, 100000)
            return
        se_case(R, "se_k2", 2, 400, 42, 1000, 200, 2, 1)
        se_case(R, "se_k2_default", 2, 300, 7, 1000, 100, 10, 6)
        se_case(R, "se_k3", 3, 300, 11, 600, 100, 5, 3, overhang=4)
        se_case(R, "se_k5", 5, 400, 13, 500, 100, 3, 2)
        se_case(R, "se_k10", 10, 500, 17, 300, 60, 7, 2)
        se_case(R, "se_k2_lagrem", 2, 100, 19, 207, 50, 10, 3)   # lag does not divide M-B (C8)
        pe_case(R, "pe_k2", 2, 300, 23, 600, 100, 2, 2)
        pe_case(R, "pe_k3", 3, 250, 29, 400, 50, 5, 3)
        pe_case(R, "pe_k5", 5, 200, 31, 300, 50, 3, 1)
        cigar_edge_case(R)
        atp2b1_case(R)
        convergent_case(R, "se_k3_convergent", False, 3, 200, 37, 50, 10, 1, 4, 700)
        convergent_case(R, "se_k2_convergent", False, 2, 300, 41, 60, 20, 2, 3, 2000)
        convergent_case(R, "pe_k2_convergent", True, 2, 200, 43, 60, 20, 2, 3, 1500)
        convergent_case(R, "pe_k4_convergent", True, 4, 200, 47, 80, 30, 2, 2, 100000)
        marginal_cases(R)
        assignment_matrix_case(R)
        as_cases(R)
    finally:
        os.dup2(saved, 1)
    print("done")


if __name__ == "__main__":
    main()
