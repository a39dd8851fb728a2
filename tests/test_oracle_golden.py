"""CPU: the oracle (oracle/miso_oracle.c, stream mode) replays the golden vectors -- outputs of
the REAL reference C core for fixed inputs and MT19937 seeds -- and must reproduce every bit:
match matrix, read classes, psi samples, log scores, final assignment, accept counts."""
import numpy as np
import pytest

import _golden
from _problems import flat


def _replay(orc, g):
    gene = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    orc.rng_seed(g["seed"])
    return gene


@pytest.mark.parametrize("name", _golden.names("se"))
def test_single_end_golden(orc, name):
    g = _golden.load(name)
    gene = _replay(orc, g)
    if "expr" in g:  # the reads themselves came from the reference simulator on the same stream
        rc, iso, pos, cig = orc.simulate_reads(gene, g["expr"], len(g["pos"]), g["read_len"])
        assert rc == 0 and (pos == g["pos"]).all() and cig == g["cigars"]
        assert (iso == g["sim_isoform"]).all()
    r = orc.miso(gene, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"],
                 lag=g["lag"], chains=g["chains"], overhang=g["overhang"])
    assert r.rc == 0
    assert np.array_equal(r.match, g["match"])
    assert np.array_equal(r.class_templates, g["class_templates"])
    assert np.array_equal(r.class_counts, g["class_counts"])
    assert np.array_equal(r.rundata, g["rundata"])
    # columns past C*floor((M-B)/lag) are never written by the reference (C8: uninitialised there)
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    assert np.array_equal(r.samples[:filled], g["samples"][:filled])
    assert np.array_equal(r.loglik[:filled], g["loglik"][:filled])
    assert np.array_equal(r.assignment, g["assignment"])


@pytest.mark.parametrize("name", _golden.names("pe"))
def test_paired_end_golden(orc, name):
    g = _golden.load(name)
    gene = _replay(orc, g)
    mean, var = float(g["mean"]), float(g["var"])
    rc, iso, pos, cig = orc.simulate_paired_reads(gene, g["expr"], len(g["pos"]) // 2,
                                                  g["read_len"], mean, var)
    assert rc == 0 and (pos == g["pos"]).all() and cig == g["cigars"]
    rc, m, fl = orc.match_iso_paired(gene, g["pos"], g["cigars"], g["read_len"], mean, var)
    assert rc == 0 and np.array_equal(m, g["match"]) and np.array_equal(fl, g["fraglen"])
    r = orc.miso_paired(gene, g["pos"], g["cigars"], g["read_len"], mean, var, iters=g["iters"],
                        burn=g["burn"], lag=g["lag"], chains=g["chains"])
    assert r.rc == 0
    assert np.array_equal(r.match, g["match"])
    assert np.array_equal(r.class_templates, g["class_templates"])
    assert np.array_equal(r.class_counts, g["class_counts"])
    assert np.array_equal(r.rundata, g["rundata"])
    assert np.array_equal(r.samples, g["samples"])
    assert np.array_equal(r.loglik, g["loglik"])
    assert np.array_equal(r.assignment, g["assignment"])


@pytest.mark.parametrize("name", _golden.names("se_conv") + _golden.names("pe_conv"))
def test_convergent_mean_golden(orc, name):
    """stop=CONVERGENT_MEAN (miso.c:556-636, 903-925, 976-983; miso_paired.c:501-523): the oracle's rounds against the
    real reference's -- the schedules are short enough that the first round does not converge, so what is compared is
    the tail of a later round of the continued chains (and, paired-end, accept counts summed over the rounds)."""
    g = _golden.load(name)
    paired = str(g["kind"]) == "pe_conv"
    kw = dict(iters=g["iters"], burn=g["burn"], lag=g["lag"], chains=g["chains"])

    def run(stop):
        gene = _replay(orc, g)
        if paired:
            mean, var = float(g["mean"]), float(g["var"])
            orc.simulate_paired_reads(gene, g["expr"], len(g["pos"]) // 2, g["read_len"], mean, var)
            return orc.miso_paired(gene, g["pos"], g["cigars"], g["read_len"], mean, var, stop=stop,
                                   max_iters=g["max_iters"], **kw)
        orc.simulate_reads(gene, g["expr"], len(g["pos"]), g["read_len"])
        return orc.miso(gene, g["pos"], g["cigars"], g["read_len"], stop=stop, max_iters=g["max_iters"], **kw)

    r = run(1)
    assert r.rc == 0
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    assert np.array_equal(r.rundata, g["rundata"])
    assert np.array_equal(r.samples[:filled], g["samples"][:filled])
    assert np.array_equal(r.loglik[:filled], g["loglik"][:filled])
    assert np.array_equal(r.assignment, g["assignment"])
    fixed = run(0)                                    # more than one round ran: not the FIXEDNO answer
    assert not np.array_equal(fixed.samples[:filled], g["samples"][:filled])


@pytest.mark.parametrize("name", _golden.names("se_marginal"))
def test_marginal_algorithm_golden(orc, name):
    """algorithm=MARGINAL (miso.c:272-283, 800-808, 936-946) against the real reference's run: samples, log scores,
    accept counts, the match matrix as the reference returns it (divided by the effective lengths) and the one
    reassignment made after the run."""
    g = _golden.load(name)
    gene = _replay(orc, g)
    orc.simulate_reads(gene, g["expr"], len(g["pos"]), g["read_len"])
    r = orc.miso(gene, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"], lag=g["lag"],
                 chains=g["chains"], overhang=g["overhang"], algo=1, stop=g["stop"], max_iters=g["max_iters"])
    assert r.rc == 0
    for f in ("match", "class_templates", "class_counts", "rundata", "assignment"):
        assert np.array_equal(getattr(r, f), g[f]), f
    assert not np.isin(g["match"], (0.0, 1.0)).all()
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    assert np.array_equal(r.samples[:filled], g["samples"][:filled])
    assert np.array_equal(r.loglik[:filled], g["loglik"][:filled])


def test_rng_stream_golden(orc):
    """MT19937 + inversion normal (random.c:301-448, 1543-1551): first draws after seeding."""
    g = _golden.load("se_k2")
    orc.rng_seed(g["seed"])
    got = np.array([orc.unif01() for _ in range(8)] + [orc.normal01() for _ in range(8)])
    assert np.array_equal(got, g["rng_stream"])


def test_cigar_edges_golden(orc):
    g = _golden.load("cigar_edges")
    gene = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    for ov in (1, 4):
        rc, m = orc.match_iso(gene, g["pos"], g["cigars"], g["read_len"], overhang=ov)
        assert rc == 0 and np.array_equal(m, g["match_ov%d" % ov]), ov


def test_lag_remainder_leaves_zero_columns(orc):
    """C8: noSamples = C*(M-B)/lag in int arithmetic, trailing sample columns stay zero."""
    g = _golden.load("se_k2_lagrem")
    S = g["chains"] * (g["iters"] - g["burn"]) // g["lag"]
    assert g["samples"].shape[0] == S == int(g["rundata"][8])
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    gene = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    orc.rng_seed(g["seed"])
    orc.simulate_reads(gene, g["expr"], len(g["pos"]), g["read_len"])
    r = orc.miso(gene, g["pos"], g["cigars"], g["read_len"], iters=g["iters"], burn=g["burn"],
                 lag=g["lag"], chains=g["chains"])
    assert filled < S and (r.samples[filled:] == 0).all() and (r.samples[:filled] != 0).all()


def test_philox_known_answers(orc):
    """include/miso_philox.h against the Random123 distribution's known-answer vectors (kat_vectors: philox4x32 at 7 and
    at 10 rounds; Salmon et al., SC'11): zero, all ones, digits of pi.  The contract draws with 7 rounds -- the fewest at
    which the paper's Table 2 lists Philox4x32 as Crush-resistant -- and the same round function reproduces the
    10-round vectors, so it is the published generator, not a look-alike."""
    inputs = [([0, 0, 0, 0], [0, 0]), ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2),
              ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])]
    kat = {7: [[0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48], [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662],
               [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]],
           10: [[0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd],
                [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]]}
    for rounds, want in kat.items():
        for (ctr, key), w in zip(inputs, want):
            assert list(orc.philox(ctr, key, rounds=rounds)) == w, (rounds, ctr)
    assert orc.philox_rounds() == 7
    for (ctr, key), w in zip(inputs, kat[7]):
        assert list(orc.philox(ctr, key)) == w


def test_two_isoform_uniforms_are_two_half_words_of_two_streams(orc):
    """The addressing of the lazy low bits as include/miso_philox.h states it, written out here from the generator alone:
    read r's uniform = (half-word r % 8 of block r // 8 at site 2) << 16 | (the same half-word at site 4), half-word h =
    bits 16 (h & 1) .. + 15 of word h // 2; counter = (block, iteration, site | chain << 8, event), key = the seed's halves."""
    seed, event, chain, iteration = 0x1234567890ABCDEF, 777, 3, 41
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for r in (0, 1, 7, 8, 9, 15, 16, 1000, 65537):
        hi = orc.philox([r // 8, iteration, 2 | (chain << 8), event], key)
        lo = orc.philox([r // 8, iteration, 4 | (chain << 8), event], key)
        h = r % 8
        want = (((int(hi[h // 2]) >> (16 * (h & 1))) & 0xFFFF) << 16) | ((int(lo[h // 2]) >> (16 * (h & 1))) & 0xFFFF)
        assert orc.split_word(seed, event, chain, iteration, r) == want, r


def test_assignment_matrix_golden(orc):
    """The gene's possible read classes (what algorithm=CLASSES sums over) against the reference's own
    splicing_assignment_matrix, stored (tests/golden/assignment_matrix.npz)."""
    z = np.load(_golden.GOLDEN_DIR + "/assignment_matrix.npz", allow_pickle=False)
    for i in range(int(z["n"])):
        exons = [tuple(int(v) for v in e) for e in z["exons_%d" % i]]
        isoforms, cur = [], []
        for v in z["isoforms_%d" % i]:
            if v < 0:
                isoforms.append(cur)
                cur = []
            else:
                cur.append(int(v))
        got = orc.assignment_matrix(orc.gene(flat(exons), isoforms), 36)
        assert np.array_equal(got, z["matrix_%d" % i]), i
