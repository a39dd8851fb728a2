"""Synthetic events shared by the tests (shapes from SURVEY.md section 8d).

Reads come from the oracle's restatement of the reference simulators (simulator.c:68-196,
221-442), seeded through the MT19937 stream, so every test sees identical inputs anywhere.
"""
import numpy as np


def se_gene(K, exlen=120, gap=100):
    """K+1 exons; isoform 0 keeps all, isoform k>0 skips exon k (K=2: the classic SE event)."""
    exons, s = [], 1
    for _ in range(K + 1):
        exons.append((s, s + exlen - 1))
        s += exlen + gap
    isoforms = [list(range(K + 1))] + [[e for e in range(K + 1) if e != k] for k in range(1, K)]
    return exons, isoforms


def flat(exons):
    return [c for e in exons for c in e]


def expr_for(K):
    w = np.arange(1, K + 1, dtype=np.float64)
    return w / w.sum()


def simulate_se(orc, K, n_reads, read_len=36, seed=7, exlen=120, gap=100):
    exons, isoforms = se_gene(K, exlen, gap)
    g = orc.gene(flat(exons), isoforms)
    orc.rng_seed(seed)
    rc, iso, pos, cig = orc.simulate_reads(g, expr_for(K), n_reads, read_len)
    assert rc == 0
    return exons, isoforms, g, pos, cig


def simulate_pe(orc, K, n_pairs, read_len=36, mean=250.0, var=900.0, seed=11, exlen=500, gap=300):
    exons, isoforms = se_gene(K, exlen, gap)
    g = orc.gene(flat(exons), isoforms)
    orc.rng_seed(seed)
    rc, iso, pos, cig = orc.simulate_paired_reads(g, expr_for(K), n_pairs, read_len, mean, var)
    assert rc == 0
    return exons, isoforms, g, pos, cig
