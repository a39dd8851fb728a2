"""GPU: events of 20 ... 60 000 reads in ONE launch, bit for bit against the oracle's counter mode.

The reference spends O(reads) per event and events share nothing (miso.c:845-900); the two-isoform kernels
therefore give every event its own number of lanes per chain -- up to a whole workgroup for the largest
(kernels_k2m.hip, plan.cpp) -- inside one launch.  Results must not depend on that choice: the same batch is
sampled with the planner's widths, with every chain squeezed onto the narrowest layout, with every chain spread
as wide as its draws allow (small bound on a wavefront's step), and with the single-width launch.
"""
import os

import numpy as np
import pytest

import miso_amd
from _libs import OrcLib
from _problems import flat, se_gene, expr_for

pytestmark = pytest.mark.gpu

SIZES = [20, 60000, 300, 5, 0, 2500, 20000, 40, 1000, 150, 7000, 64, 33000, 3, 511]


def _env(**kw):
    class _Ctx:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            for k, v in kw.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _Ctx()


def _events(orc, paired, sizes):
    out = []
    for j, n in enumerate(sizes):
        exons, isoforms = se_gene(2, exlen=500 if paired else 90 + 7 * j, gap=300 if paired else 100)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(4000 + j)
        w = np.array([0.2 + 0.05 * (j % 12), 1.0])
        if paired:
            rc, _, pos, cig = orc.simulate_paired_reads(g, w / w.sum(), max(n, 1), 36, 250.0, 900.0)
            pos, cig = pos[:2 * n], cig[:2 * n]
        else:
            rc, _, pos, cig = orc.simulate_reads(g, w / w.sum(), max(n, 1), 36)
            pos, cig = pos[:n], cig[:n]
        assert rc == 0
        out.append((exons, isoforms, g, pos, cig))
    return out


@pytest.mark.parametrize("paired,chains", [(False, 1), (False, 3), (True, 1), (True, 2)])
def test_mixed_read_counts_one_launch_bit_exact(orc, paired, chains):
    sizes = SIZES if not paired else [s // 2 for s in SIZES]
    evs = _events(orc, paired, sizes)
    kw = dict(iters=90, burn=20, lag=3, chains=chains)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        if paired:
            r = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=11, event_id=700 + i,
                                trace=True, **kw)
        else:
            r = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=11, event_id=700 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    variants = [dict(), dict(MISO_K2_TARGET="1e12"), dict(MISO_K2_TARGET="1400"), dict(MISO_K2_TARGET="2600"),
                dict(MISO_K2_MULTI="0")]
    # chains on several workgroups (coop.hpp): a small bound and one quad per lane put the largest events on 2 ... 8
    variants += [dict(MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1"), dict(MISO_K2_TARGET="1400", MISO_NO_COOP="1")]
    if not paired:   # one round: the wavefronts paired by estimated duration across the runs (a table entry per wavefront) or within
        variants += [dict(MISO_K2_GLOBAL_PAIR="1"), dict(MISO_K2_GLOBAL_PAIR="0"), dict(MISO_K2_GLOBAL_PAIR="1", MISO_K2_TARGET="2600"),
                     dict(MISO_K2_GLOBAL_PAIR="1", MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1")]
    if paired:   # every event through MODE 1 (the kernel for events with non-finite scores) and its many-widths launch
        variants += [dict(MISO_NO_PE_DELTA="1"), dict(MISO_NO_PE_DELTA="1", MISO_K2_TARGET="1400"), dict(MISO_K2W_WPB="8"),
                     dict(MISO_NO_PE_DELTA="1", MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1"),
                     dict(MISO_K2W_WPB="8", MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1")]
    kernels = set()
    for v in variants:
        with _env(**v):
            b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0,
                               device_match=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=11, first_event_id=700)
            kernels.add(b.last_kernels())
            for i, r in enumerate(cpu):
                gpu = b.result(i)
                where = (v, sizes[i], b.last_kernels())
                assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
                assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
                assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
                assert (gpu.assignment == r.assignment).all(), where
                assert gpu.rundata.noAccepted == r.accepted, where
    assert any("multi" in k for k in kernels), kernels


@pytest.mark.parametrize("K,chains", [(3, 1), (5, 2), (10, 1), (18, 1)])
def test_mixed_read_counts_three_or_more_isoforms_bit_exact(orc, K, chains):
    """sampler_flat: wavefronts packed by work units, the largest chains one per workgroup (runtime.hip flat_waves,
    kernels_flat.inl FLAT_WIDE) against the uniform layout and the oracle."""
    sizes = [20, 40000, 300, 5, 0, 2500, 12000, 40, 1000, 150, 7000, 64, 3, 511, 90, 200, 35, 800]
    evs = []
    for j, n in enumerate(sizes):
        exons, isoforms = se_gene(K, exlen=90 + 7 * j, gap=100)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(5000 + j)
        rc, _, pos, cig = orc.simulate_reads(g, expr_for(K), max(n, 1), 36)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:n], cig[:n]))
    kw = dict(iters=60, burn=10, lag=2, chains=chains)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=13, event_id=900 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    variants = [dict(), dict(MISO_FLAT_PACK="1"), dict(MISO_FLAT_PACK="1", MISO_FLAT_NC="4"),
                dict(MISO_FLAT_PACK="0", MISO_FLAT_NC="3"), dict(MISO_FLAT_PACK="1", MISO_FLAT_NC="2", MISO_FLAT_NO_DESC="1")]
    for v in variants:
        with _env(**v):
            b = miso_amd.Batch(36, device_match=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=13, first_event_id=900)
            assert "sampler_flat" in b.last_kernels()
            for i, r in enumerate(cpu):
                gpu = b.result(i)
                where = (v, K, sizes[i])
                assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
                assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
                assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
                assert (gpu.assignment == r.assignment).all(), where
                assert gpu.rundata.noAccepted == r.accepted, where


@pytest.mark.parametrize("sd,forced", [(30.0, True), (450.0, False)])
def test_two_isoform_paired_end_through_the_general_kernel(orc, sd, forced):
    """Fragment-length distributions too wide for the two-isoform kernel's LDS tables (sd beyond ~400; the
    reference takes any sd, miso_paired.c:299-308) send K = 2 paired-end events to the general kernel instead of
    failing the batch; MISO_K2_GENERAL=1 forces that route at sd = 30.  Bit-exact against the oracle either way."""
    mean, var = 250.0 if sd < 100 else 1500.0, sd * sd
    evs = []
    for j, n in enumerate([300, 40, 900, 0, 150]):
        exons, isoforms = se_gene(2, exlen=600 if sd < 100 else 4000, gap=300)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(6000 + j)
        rc, _, pos, cig = orc.simulate_paired_reads(g, np.array([0.4, 0.6]), max(n, 1), 36, mean, var)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:2 * n], cig[:2 * n]))
    kw = dict(iters=80, burn=20, lag=2, chains=2)
    with _env(MISO_K2_GENERAL="1" if forced else None):
        b = miso_amd.Batch(36, paired=True, mean=mean, var=var, device_match=True, **kw)
        for exons, isoforms, g, pos, cig in evs:
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        b.run(seed=21, first_event_id=40)
    assert "sampler_k2" not in b.last_kernels(), b.last_kernels()
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso_paired(g, pos, cig, 36, mean, var, mode=OrcLib.COUNTER, seed=21, event_id=40 + i, trace=True, **kw)
        assert r.rc == 0
        gpu = b.result(i)
        assert (gpu.counts_hash == r.trace["counts_hash"]).all(), (sd, i)
        assert np.array_equal(gpu.samples, r.samples, equal_nan=True), (sd, i)
        assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), (sd, i)
        assert (gpu.assignment == r.assignment).all(), (sd, i)


@pytest.mark.parametrize("K,chains", [(3, 2), (5, 1), (10, 1)])
def test_mixed_pair_counts_three_or_more_isoforms_paired_end_bit_exact(orc, K, chains):
    """sampler_grp, paired-end: genes of very different sizes are split into size buckets per isoform-count class
    (runtime.hip upload) -- the mean-sized on the rule's lanes per chain, the several-times-larger on at least 32, the
    next a WAVEFRONT each (sampler_grp<64, true, KC>), the largest one per WORKGROUP (kernels_grp.inl WIDE: 256 lanes,
    totals through LDS) -- launched side by side.
    Against the single launch per class (MISO_NO_PE_BUCKETS=1) and the oracle."""
    sizes = [30, 22000, 200, 5, 0, 1800, 9000, 60, 700, 120, 90, 200, 35, 400, 150, 80]
    evs = []
    for j, n in enumerate(sizes):
        exons, isoforms = se_gene(K, exlen=500 + 11 * j, gap=300)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(7000 + j)
        rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(K), max(n, 1), 36, 250.0, 900.0)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:2 * n], cig[:2 * n]))
    kw = dict(iters=50, burn=10, lag=2, chains=chains)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=17, event_id=300 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    names = []
    # (MISO_COOP_DRAWS: drawing pairs per workgroup of a chain on SEVERAL workgroups, coop.hpp: 1024 puts the two largest
    # genes on ~10 and ~4 workgroups; MISO_NO_COOP: one workgroup per chain)
    for v in (dict(), dict(MISO_NO_PE_BUCKETS="1"), dict(MISO_PE_FORCE_EXACT="1"), dict(MISO_COOP_DRAWS="1024"),
              dict(MISO_NO_COOP="1"), dict(MISO_PE_T_WAVE="1", MISO_PE_T_WIDE="1e9"),
              dict(MISO_PE_T_WAVE="64", MISO_PE_T_WIDE="1500"), dict(MISO_PE_MULTI="1"),
              dict(MISO_PE_MULTI="1", MISO_PE_T_WAVE="1", MISO_PE_T_WIDE="1e9"), dict(MISO_PE_MULTI="1", MISO_PE_T_WAVE="64", MISO_PE_T_WIDE="1500"),
              dict(MISO_PE_MULTI="1", MISO_PE_T_WAVE="64", MISO_PE_T_WIDE="1500", MISO_COOP_DRAWS="1024")):
        with _env(**v):
            b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, device_match=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=17, first_event_id=300)
            names.append(b.last_kernels())
            for i, r in enumerate(cpu):
                gpu = b.result(i)
                where = (v, K, sizes[i], b.last_kernels())
                assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
                assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
                assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
                assert (gpu.assignment == r.assignment).all(), where
                assert gpu.rundata.noAccepted == r.accepted, where
    assert ", true>" in names[0] and names[0].count("sampler_grp") >= 3, names     # workgroup-wide + 32-lane + normal launches
    # MISO_PE_MULTI=1 (what a batch of many classes does by itself): the class's size buckets in ONE launch (sampler_grp_multi)
    assert all(n == "sampler_grp_multi<%d>" % {3: 4, 5: 8, 10: 12}[K] for n in names[7:]), names
    assert names[1].count("sampler_grp") == 1, names
    kc = {3: 4, 5: 8, 10: 12}[K]
    assert "sampler_grp<64, true, %d>" % kc in names[5] and ", true>" not in names[5], names   # a wavefront per large gene, none workgroup-wide
    assert "sampler_grp<64, true, %d>" % kc in names[6] and "sampler_grp<64, true, %d, true>" % kc in names[6], names


def test_whole_gene_batch_of_several_classes_in_one_launch_bit_exact(orc):
    """sampler_grp_all (kernels_grp.inl): a paired-end batch of like-sized genes of several isoform-count classes (sixteen lanes
    per chain everywhere) is ONE launch, its classes segments ordered by cost.  Against the oracle, with every read through the
    exact scan, and against the launch per class (MISO_NO_PE_ALL=1), which must give the same bits; a batch with size buckets
    (one gene of 9000 pairs among them) keeps the launches per class."""
    counts = [3, 18, 5, 10, 14, 4, 7, 12, 20, 9, 3, 5, 16, 8, 11, 6, 13, 17, 19, 15]
    sizes = [300, 260, 200, 350, 220, 180, 0, 260, 400, 320, 290, 200, 235, 400, 250, 380, 310, 270, 330, 240]
    evs = []
    for j, (K, n) in enumerate(zip(counts, sizes)):
        exons, isoforms = se_gene(K, exlen=500 + 11 * j, gap=300)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(7100 + j)
        rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(K), max(n, 1), 36, 250.0, 900.0)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:2 * n], cig[:2 * n]))
    kw = dict(iters=40, burn=10, lag=2, chains=2)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=23, event_id=900 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    names = []
    # (a batch this small would get 32 lanes per chain and size buckets: sixteen everywhere is what 16 384 genes get by themselves)
    lanes16 = dict(MISO_GENERAL_LANES="16", MISO_NO_PE_BUCKETS="1", MISO_PE_ALL="1")
    for v in (lanes16, dict(lanes16, MISO_PE_FORCE_EXACT="1"), dict(lanes16, MISO_NO_PE_ALL="1")):
        with _env(**v):
            b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, device_match=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=23, first_event_id=900)
            names.append(b.last_kernels())
            b.run(seed=23, first_event_id=900)   # (the segment table is kept between launches)
            assert b.last_kernels() == names[-1]
            for i, r in enumerate(cpu):
                gpu = b.result(i)
                where = (v, counts[i], sizes[i], b.last_kernels())
                assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
                assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
                assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
                assert (gpu.assignment == r.assignment).all(), where
                assert gpu.rundata.noAccepted == r.accepted, where
    assert names[0] == "sampler_grp_all" and names[1] == "sampler_grp_all", names
    assert "sampler_grp_all" not in names[2] and names[2].count("sampler_grp<16, true, ") == 5, names
    # one large gene: size buckets, a launch per class (or per run) as before
    exons, isoforms = se_gene(6, exlen=700, gap=300)
    g = orc.gene(flat(exons), isoforms)
    orc.rng_seed(7177)
    rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(6), 9000, 36, 250.0, 900.0)
    assert rc == 0
    b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, device_match=True, **kw)
    for exons_, isoforms_, g_, pos_, cig_ in evs:
        b.add_event(miso_amd.Gene(exons_, isoforms_), pos_, cig_)
    b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=23, first_event_id=900)
    assert "sampler_grp_all" not in b.last_kernels(), b.last_kernels()
    for i, r in enumerate(cpu):
        assert np.array_equal(b.result(i).samples, r.samples, equal_nan=True), i


def test_small_genes_of_a_whole_gene_batch_on_eight_lanes_bit_exact(orc):
    """runtime.hip upload, bucket -1: in a paired-end batch of several isoform-count classes the genes of few pairs (up to
    MISO_PE_T_SMALL drawing quads, default 96) run on EIGHT lanes per chain, eight chains of a wavefront sharing the scalar step,
    their score tables in global memory -- a segment of the class's sampler_grp_multi launch with a tstride of its own.  Against
    the oracle with the bucket on (two thresholds), off, and with every read through the exact scan."""
    counts = [3, 18, 5, 10, 14, 4, 7, 12, 20, 9, 3, 5, 16, 8, 11, 6, 5, 5, 9, 13]
    sizes = [30, 60, 200, 5, 120, 1800, 0, 60, 700, 90, 90, 200, 35, 400, 150, 80, 40, 45, 55, 65]
    evs = []
    for j, (K, n) in enumerate(zip(counts, sizes)):
        exons, isoforms = se_gene(K, exlen=500 + 11 * j, gap=300)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(7300 + j)
        rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(K), max(n, 1), 36, 250.0, 900.0)
        assert rc == 0
        evs.append((exons, isoforms, g, pos[:2 * n], cig[:2 * n]))
    kw = dict(iters=40, burn=10, lag=2, chains=2)
    cpu = []
    for i, (exons, isoforms, g, pos, cig) in enumerate(evs):
        r = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=29, event_id=500 + i, trace=True, **kw)
        assert r.rc == 0
        cpu.append(r)
    names = []
    for v in (dict(), dict(MISO_PE_T_SMALL="20"), dict(MISO_PE_T_SMALL="0"), dict(MISO_PE_FORCE_EXACT="1"), dict(MISO_NO_PE_MULTI="1")):
        with _env(**v):
            b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, device_match=True, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            b.run(seed=29, first_event_id=500)
            names.append(b.last_kernels())
            for i, r in enumerate(cpu):
                gpu = b.result(i)
                where = (v, counts[i], sizes[i], b.last_kernels())
                assert (gpu.counts_hash == r.trace["counts_hash"]).all(), where
                assert np.array_equal(gpu.samples, r.samples, equal_nan=True), where
                assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), where
                assert (gpu.assignment == r.assignment).all(), where
                assert gpu.rundata.noAccepted == r.accepted, where
    assert "sampler_grp_multi<" in names[0], names
    assert "sampler_grp<8, true, " in names[4], names   # without the multi kernel the small genes' runs are eight-lane launches of their own
    # (round 6: the launch's statistics have a record per run -- more than the sixteen the binding used to ask for -- and cover every chain)
    st = b.launch_stats()
    assert sum(k["chains"] for k in st["kernels"]) == len(evs) * kw["chains"], st


def _worker(paired, K, rounds, **env):
    import subprocess
    import sys
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    return subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_coop_worker.py"),
                             str(int(paired)), str(K), str(rounds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


@pytest.mark.parametrize("paired,K,env", [
    (True, 5, dict(MISO_COOP_DRAWS="1024")),                                   # sampler_grp WIDE, ~10 and ~4 workgroups per chain
    (True, 5, dict(MISO_COOP_DRAWS="1024", MISO_PE_MULTI="1")),                # the same inside sampler_grp_multi
    (False, 2, dict(MISO_K2_TARGET="1000", MISO_COOP_MIN_QUADS="1")),          # sampler_k2_multi<0, 8>
    (True, 2, dict(MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1")),           # sampler_k2_multi<2, 4>
])
def test_a_chain_that_times_out_on_its_workgroups_is_rerun_not_failed(paired, K, env):
    """coop.hpp's second line of defence, forced: with MISO_COOP_MAX_POLLS=1 a cooperative workgroup gives up on its
    chain's other workgroups after ONE poll (as on a device too busy to make them resident); every workgroup of the
    chain leaves, miso_batch_sync() repeats the launch in the same process with one workgroup per chain and the caller
    gets the oracle's results bit for bit -- round 3 failed the WHOLE batch with MISO_EINTERNAL here.  The reference's
    workers share nothing (misopy/miso.py:165-187): one gene must never cost a batch its results."""
    p = _worker(paired, K, 1, MISO_COOP_MAX_POLLS="1", **env)
    out, err = p.communicate(timeout=900)
    assert p.returncode == 0, out + err
    assert out.startswith("ok retries=") and int(out.split("retries=")[1].split()[0]) >= 1, out + err
    assert "re-running the launch with one workgroup per chain" in err


def test_two_processes_on_one_gpu_with_cooperative_chains_in_both():
    """`miso -p N` with more workers than GPUs, or any second tenant: processes sample heavy-tailed batches on ONE device
    at the same time, all with chains on several workgroups in flight (paired-end genes through sampler_grp's WIDE path,
    two-isoform events through sampler_k2_multi), several launches each.  All must return the oracle's results bit for
    bit; a time-out, should the device ever be that busy, is absorbed by the re-run (miso_batch_coop_retries)."""
    procs = [_worker(True, 5, 6, MISO_COOP_DRAWS="1024"), _worker(False, 2, 6, MISO_K2_TARGET="1000", MISO_COOP_MIN_QUADS="1"),
             _worker(True, 2, 6, MISO_K2_TARGET="1400", MISO_COOP_MIN_QUADS="1")]
    for p in procs:
        out, err = p.communicate(timeout=1500)
        assert p.returncode == 0, out + err
        assert out.startswith("ok retries="), out + err


def test_collapsed_batch_on_one_device_then_another():
    """release() between two uploads must leave nothing behind (the log-factorial table of the collapsed step, the
    cross-run pairing table and the cached plans were kept as stale pointers in round 3: use-after-free on the second
    device).  Needs two GPUs."""
    from miso_amd import capi
    if capi.device_count() < 2:
        pytest.skip("one GPU visible")
    orc = OrcLib()
    evs = _events(orc, False, [300, 5000, 40, 900, 64, 2500, 150, 700])
    kw = dict(iters=80, burn=20, lag=2, chains=2)
    for collapsed, env in ((True, {}), (False, dict(MISO_K2_GLOBAL_PAIR="1"))):
        with _env(**env):
            b = miso_amd.Batch(36, device_match=True, collapsed=collapsed, **kw)
            for exons, isoforms, g, pos, cig in evs:
                b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            res = []
            for dev in (0, 1, 0):
                b.run(device=dev, seed=5, first_event_id=10)
                res.append([b.result(i) for i in range(len(evs))])
            for i in range(len(evs)):
                assert np.array_equal(res[0][i].samples, res[1][i].samples) and np.array_equal(res[0][i].samples, res[2][i].samples)
                assert (res[0][i].counts_hash == res[1][i].counts_hash).all()
