#!/usr/bin/env python3
"""bench.py -- AS events/sec of the MI355X MISO sampler on BASELINE.json's metric shape.

One "step" = one pass of the hot path (all chains, all iterations) over one batch of synthetic
events already resident in HBM.  Headline workload (BASELINE.json configs[1]): skipped-exon events,
2 isoforms, 1000 single-end 36-bp reads, 2500 burn-in + 5000 kept iterations, lag 1, 1 chain.

    python bench.py [--gpus N --steps K --warmup W] [--events E per GPU] [--K 2] [--reads 1000]

`--gpus N` (N > 1) starts N ranks itself -- one fresh process per GPU through
`python -m torch.distributed.run`, before anything in this process touches the GPU (the reference's
dispatcher starts its own workers too, misopy/miso.py:165-187, 205-214) -- unless it is already
running under a launcher (WORLD_SIZE set).  Rank 0 prints ONE compact JSON line (< 4 KB: the driver keeps the tail of
stdout) as the LAST line of stdout and writes the full record -- every note, model detail and per-row roofline -- to
gpurun_out/bench_full.json (`--full-out`).  The global event list
(N x --events events) is split into N contiguous shards balanced by cost (sum of
chains x iterations x reads), every event keeps its global id, there is no collective on the data
path; ranks only meet in a gloo barrier and a max-over-ranks of the elapsed time.

The line carries, besides the contract's keys:
  roofline            the dominant kernel against the bound that applies to it: VALU issue (VALU
                      instructions per chain-iteration and their issue cycles from the committed
                      rocprofv3 passes of this workload, profiles/valu_model.json -- a MODEL priced with
                      this run's kernel time, and said so), with the share of the instruction floor,
                      the RNG fraction, the measured HBM fraction and SURVEY 8(d)'s algorithmic bytes
  cpu_baseline        the real reference C core (oracle/_ref) on the host cores, bounded sample
  delta_psi           |delta psi| with a pass/fail: the GPU under S streams per event against the real reference under S
                      streams per event, exact permutation tests (tests/_dpsi.py): pooled t^2, largest |t|, signed shifts
  summary_ms / compare_ms  rows f2 / f3 (configs[4]) on the resident samples, outside the timed region
  matrix              the other shapes of the metric (hg19-like read counts; K = 5, 10; MISO default
                      settings; paired-end K = 2, 5, 10; the whole-gene paired-end mix), each timed in
                      this run with its own roofline, cpu_baseline and delta_psi (all isoforms, mean and
                      both credible-interval bounds)
"""
import argparse
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything initialises HIP (torch in the multi-GPU runs): docs/history.md 4.3 (iv)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SIMDS = 256 * 4              # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # max shader clock (MI355X_MICROARCH.md)
VALU_PEAK_GCYC = SIMDS * CLOCK_GHZ   # VALU issue cycles per ns of wall time, whole chip
PHILOX_ROUNDS = 7            # include/miso_philox.h MISO_PHILOX_ROUNDS
# Every row's |delta psi| test reports pass / fail at its level of 1e-3 in the line.  A row that fails is tested ONCE more
# on fresh random streams on BOTH sides (new reference seeds, new Philox seed: `retry`); the PROCESS fails (exit 3) when
# any single row -- the headline or a matrix row -- fails both times.  (Round 4 asked for two distinct failing rows, which
# let a defect in a kernel that serves one row only -- sampler_flat<12>, sampler_grp<16, true, 12> -- through; eighteen
# honest rows at 1e-3 each fail one of them once in sixty runs, hence the second look instead of failing at once: a false
# alarm needs two independent failures, 1e-6 per row, a real defect fails again.)
RETRY_SEED_SHIFT = 7919


# ------------------------------------------------------------------------------------------------
# N ranks from one command
# ------------------------------------------------------------------------------------------------
def self_launch(n_gpus):
    """Re-run this command as n_gpus ranks (children of this process; nothing here has touched the
    GPU yet) and return their exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# ------------------------------------------------------------------------------------------------
# workloads: the headline and the matrix rows share one description
# ------------------------------------------------------------------------------------------------
BASE_SHAPE = dict(K=2, reads=1000, read_len=36, iters=7500, burn=2500, lag=1, chains=1, paired=False, mean=250.0, var=900.0)


def shape_of(a, **ov):
    """{K, reads ("hg19" or a number), read_len, iters, burn, lag, chains, paired, mean, var} of a workload."""
    sh = dict(K=a.K, reads=("hg19" if a.reads_dist == "hg19" else a.reads), read_len=a.read_len, iters=a.iters,
              burn=a.burn, lag=a.lag, chains=a.chains, paired=bool(a.paired), mean=250.0, var=900.0)
    sh.update(ov)
    if isinstance(sh["K"], list):
        sh["K"] = tuple(sh["K"])
    return sh


def reads_spec(sh):
    from miso_amd import workload
    return workload.HG19_LIKE if sh["reads"] == "hg19" else sh["reads"]


def workload_key(events, sh):
    """One canonical key per workload (profiles/valu_model.json, traffic.json): K as "lo-hi" for isoform
    ranges, reads as a number or "hg19".  The kernel is NOT part of the key: the entry names the kernels it
    was profiled with and bench.py refuses a profile of other kernels."""
    K = sh["K"]
    k = "%d-%d" % tuple(K) if isinstance(K, (tuple, list)) else str(K)
    return "events=%d|K=%s|reads=%s|iters=%d|chains=%d|paired=%d" % (
        events, k, sh["reads"], sh["iters"], sh["chains"], int(sh["paired"]))


def build(first, n, sh, device_match=True, collapsed=False):
    from miso_amd import workload
    return workload.build_batch(first, n, K=sh["K"], n_reads=reads_spec(sh), read_len=sh["read_len"],
                                iters=sh["iters"], burn=sh["burn"], lag=sh["lag"], chains=sh["chains"],
                                paired=sh["paired"], mean=sh["mean"], var=sh["var"], device_match=device_match,
                                collapsed=collapsed)


MATRIX = [  # (id, label, shape overrides, events, reference runs of the row: (events, seeds) or None)
    # read counts as a real annotation sees them (workload.HG19_LIKE: log-normal, median 300, 20 ... 10^5 reads per
    # event, a handful of 10^4 ... 10^5-read events per 40 000): "reads_iter_per_s" is what to compare with the uniform rows
    ("se_k2_hg19", "SE K=2, hg19-like read counts (20..1e5 per event), 1 chain, 7500 iters", dict(reads="hg19"), 40000, (64, 8)),
    ("se_k2_hg19_defaults", "SE K=2, hg19-like read counts, MISO defaults (6 chains, 5000 iters, 500 burn-in, lag 10)",
     dict(reads="hg19", chains=6, iters=5000, burn=500, lag=10), 40000, (64, 8)),
    ("pe_k2_hg19", "PE K=2, hg19-like read counts, 1 chain, 7500 iters", dict(paired=True, reads="hg19"), 40000, (64, 8)),
    ("se_k5_hg19", "SE K=5, hg19-like read counts, 1 chain, 7500 iters", dict(K=5, reads="hg19"), 40000, (64, 8)),
    ("pe_k5_hg19", "PE K=5, hg19-like read counts, 1 chain, 7500 iters", dict(K=5, paired=True, reads="hg19"), 40000, (64, 8)),
    ("se_k5", "SE K=5, 1 chain, 7500 iters", dict(K=5), 40000, (64, 8)),
    ("se_k10", "SE K=10, 1 chain, 7500 iters", dict(K=10), 40000, (64, 8)),
    ("se_k2_defaults", "SE K=2, MISO defaults (6 chains, 5000 iters, 500 burn-in, lag 10)",
     dict(chains=6, iters=5000, burn=500, lag=10), 40000, (64, 8)),
    ("pe_k2", "PE K=2 (mean 250, sd 30), 1 chain, 7500 iters", dict(paired=True), 40000, (64, 8)),
    ("pe_k5", "PE K=5, 1 chain, 7500 iters", dict(K=5, paired=True), 40000, (64, 8)),
    ("pe_k10", "PE K=10, 1 chain, 7500 iters", dict(K=10, paired=True), 20000, (64, 8)),
    # BASELINE configs[3]: whole-gene mode, 3-20 isoforms per gene, paired-end; "events" are genes here
    ("pe_mix", "PE K=3..20 per gene (whole-gene mix), 1 chain, 7500 iters", dict(K=(3, 20), paired=True), 16384, (48, 8)),
    ("pe_mix_hg19", "PE K=3..20 per gene, hg19-like read counts, 1 chain, 7500 iters", dict(K=(3, 20), paired=True, reads="hg19"), 16384, (48, 8)),
]


# ------------------------------------------------------------------------------------------------
# CPU side: the real reference (oracle/_ref) or the oracle port, timed and used for |delta psi|
# ------------------------------------------------------------------------------------------------
def _psi_stats(smp):
    """Per isoform: posterior mean, Chen-Shao 95 % bounds (credible_intervals.py:31-55), sd."""
    import numpy as np
    x = np.sort(np.asarray(smp), axis=0)
    n = x.shape[0]
    lo, hi = int(round(0.025 * n)) - 1, int(round(0.975 * n)) - 1
    return x.mean(0).tolist(), x[lo].tolist(), x[hi].tolist(), x.std(0, ddof=1).tolist()


def _cpu_worker(job):
    """One host process, one workload: `ev_seeds` = [(event id, reference seed)], run one after the other."""
    kind, sh, ev_seeds = job
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)  # the reference prints "no chains: %d" per call (miso.c:837)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import OrcLib, RefLib
    from miso_amd import workload
    L = RefLib() if kind == "reference" else OrcLib()
    probs = {}
    for e, _ in ev_seeds:
        if e not in probs:
            exons, isoforms, pos, cig = workload.event_reads(e, sh["K"], reads_spec(sh), sh["read_len"], sh["paired"],
                                                             sh["mean"], sh["var"])
            probs[e] = (L.gene([c for ex in exons for c in ex], isoforms), pos, cig)
    out = []
    busy = 0.0
    kw = dict(iters=sh["iters"], burn=sh["burn"], lag=sh["lag"], chains=sh["chains"])
    for e, seed in ev_seeds:
        g, pos, cig = probs[e]
        L.rng_seed(seed)   # the reference's generator is one global stream (random.c:491)
        t0 = time.perf_counter()
        if sh["paired"]:
            r = L.miso_paired(g, pos, cig, sh["read_len"], sh["mean"], sh["var"], **kw)
        else:
            r = L.miso(g, pos, cig, sh["read_len"], **kw)
        busy += time.perf_counter() - t0
        assert r.rc == 0
        out.append((e, seed) + _psi_stats(r.samples))
    return busy, out


def cpu_studies(wanted, start="fork"):
    """Everything the host cores do, before anything touches the GPU (fork-safe), in ONE process pool (the
    reference's own parallelism is processes, misopy/miso.py:165-187).  `wanted`: {id: (shape, timed
    [(event, seed)], study [(event, seed)])}.  Timed runs give the cpu_baseline (runs / the slowest process's
    busy time); study runs (several reference seeds per event) give the Monte-Carlo standard error of one run's
    posterior summaries, event by event (tolerance 4 x MCSE from >= 8 reference seeds).
    A workload whose timed list is empty is timed on its study runs."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import RefLib
    kind = "reference" if RefLib.available() else "port"
    cores = usable_cores()
    jobs, tags = [], []
    for wid, (sh, timed, study) in wanted.items():
        for which, lst in (("timed", timed), ("study", study)):
            for p in range(cores):
                part = lst[p::cores]
                if part:
                    jobs.append((kind, sh, part)); tags.append((wid, which))
    out = {wid: {"timed": [], "study": [], "timed_busy": [], "study_busy": []} for wid in wanted}
    t0 = time.perf_counter()
    with mp.get_context(start).Pool(cores) as pool:
        for (wid, which), (busy, rows) in zip(tags, pool.map(_cpu_worker, jobs, chunksize=1)):
            out[wid][which] += rows
            out[wid][which + "_busy"].append(busy)
    wall = time.perf_counter() - t0
    for wid, (sh, timed, study) in wanted.items():
        o = out[wid]
        which = "timed" if timed else "study"
        runs, busy = len(o[which]), o[which + "_busy"]
        n_ev = len(set(r[0] for r in o[which]))
        o["baseline"] = {
            "value": round(runs / max(busy), 3), "unit": "events/s", "cores": len(busy), "kind": kind,
            "sample": "%d reference runs (%d events x %d seed(s), the workload's own synthetic events: K=%s reads=%s iters=%d "
                      "burn=%d lag=%d chains=%d%s) on %d processes, slowest process busy %.2fs%s"
                      % (runs, n_ev, max(1, runs // max(1, n_ev)), sh["K"], sh["reads"], sh["iters"], sh["burn"], sh["lag"],
                         sh["chains"], " paired-end" if sh["paired"] else "", len(busy), max(busy),
                         "" if timed else "; the runs of the |dpsi| study are the timing sample"),
            "value_1core": round(runs / sum(busy), 3)}
    return out, wall


def gpu_streams(sh, events, n_streams, device, seed, collapsed=False):
    """The build's side of the two-sample |delta psi| test: every study event under `n_streams` independent random
    streams in ONE launch -- n_streams copies of the event (same reads: a pure function of its id) whose ids in the Philox
    counter differ (miso_batch_set_event_id; the counter's event word is what separates two events' streams,
    include/miso_philox.h) -- summarised on the device.  Returns {event: [(mean, ci_low, ci_high), ...]} and the launch's
    kernels."""
    from miso_amd import capi, workload
    kw = dict(min_len=400, max_len=800, gap=300) if sh["paired"] else {}
    b = capi.Batch(sh["read_len"], iters=sh["iters"], burn=sh["burn"], lag=sh["lag"], chains=sh["chains"], paired=sh["paired"],
                   mean=sh["mean"] if sh["paired"] else 0.0, var=sh["var"] if sh["paired"] else 0.0, device_match=True,
                   collapsed=collapsed)
    spec = reads_spec(sh)
    where = []
    for s in range(n_streams):
        for e in events:
            exons, isoforms, expr = workload.event_gene(e, workload.mixed_k(e, sh["K"]), **kw)
            i = b.add_simulated(capi.Gene(exons, isoforms), expr, workload.event_n_reads(e, spec), workload.GEN_SEED + e)
            b.set_event_id(i, ((s + 1) << 20) + e)
            where.append((e, i))
    b.upload(device)
    b.launch(seed=seed, first_event_id=0)
    b.sync()
    b.summarize(0.95)
    runs = {}
    for e, i in where:
        runs.setdefault(e, []).append(tuple(v.tolist() for v in b.summary(i)))
    return runs, b.last_kernels()


def delta_psi(sh, study_rows, device, seed, collapsed=False, n_streams=None):
    """BASELINE's "|delta psi| vs ref" with a pass/fail: the build under S random streams per event against the REAL
    reference C core under S streams per event (posterior mean and both Chen-Shao bounds of every isoform), compared by
    the exact permutation tests of tests/_dpsi.py -- pooled t^2, largest |t|, signed shifts of the mean and of either
    bound, shrinkage towards the uniform vector, interval width.  (Round 3 compared one GPU run with S reference runs cell
    by cell: little power, blind to a systematic shift.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dpsi
    ref = {}
    for e, _, m, lo, hi, _sd in study_rows:
        ref.setdefault(e, []).append((m, lo, hi))
    events = sorted(ref)
    S = min(len(v) for v in ref.values())
    if not events or S < 2:
        return None
    t0 = time.perf_counter()
    runs, kernels = gpu_streams(sh, events, n_streams or S, device, seed, collapsed)
    t_gpu = time.perf_counter() - t0
    kmax = max(len(ref[e][0][0]) for e in events)
    out = _dpsi.two_sample(_dpsi.stack_runs(runs, events, kmax), _dpsi.stack_runs(ref, events, kmax))
    out["build_side"] = "%d events x %d streams in one launch (%s), %.2f s incl. host set-up" % (
        len(events), n_streams or S, kernels, t_gpu)
    out["test_s"] = round(time.perf_counter() - t0 - t_gpu, 2)
    return out


def delta_psi_with_retry(wid, sh, study_rows, device, seed, collapsed=False):
    """The row's two-sample test; a failure is looked at once more with fresh streams on both sides (the reference runs in
    a `spawn` pool: this process has touched the GPU by now, a fork would carry its HIP state along).  Returns the first
    test's record with `retry` (the second test's) and `confirmed_fail` in it."""
    d = delta_psi(sh, study_rows, device, seed, collapsed)
    if d is None or d.get("pass") is not False:
        return d
    ev_seeds = sorted(set((r[0], r[1]) for r in study_rows))
    fresh = [(e, sd + RETRY_SEED_SHIFT * 1000003) for e, sd in ev_seeds]
    try:
        st, _ = cpu_studies({wid: (sh, [], fresh)}, start="spawn")
        d2 = delta_psi(sh, st[wid]["study"], device, seed + RETRY_SEED_SHIFT, collapsed)
    except Exception as err:   # the retry is evidence, not the measurement: a failure to run it leaves the first verdict
        d["retry"] = {"error": repr(err)[:200]}
        d["confirmed_fail"] = True
        return d
    d["retry"] = None if d2 is None else {k: d2.get(k) for k in ("pass", "p_row", "p_pooled", "p_sign", "max_z", "n_fail", "tests")}
    d["confirmed_fail"] = bool(d2 is not None and d2.get("pass") is False)
    return d


def single_run_deltas(batch, sample_rows):
    """Plain differences between the timed batch's own run and single runs of the reference (the timing sample), isoform 0."""
    import numpy as np
    rows = [r for r in sample_rows if r[0] < len(batch)]
    if not rows:
        return {}
    batch.summarize(0.95)
    d = np.array([[abs(batch.summary(r[0])[j][0] - r[2 + j][0]) for j in range(3)] for r in rows])
    return {"events": len(rows), "mean_abs_dpsi": round(float(d[:, 0].mean()), 6), "max_abs_dpsi": round(float(d[:, 0].max()), 6),
            "mean_abs_dci_low": round(float(d[:, 1].mean()), 6), "mean_abs_dci_high": round(float(d[:, 2].mean()), 6)}


# ------------------------------------------------------------------------------------------------
# roofline of one timed batch
# ------------------------------------------------------------------------------------------------
def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}


def valu_floor(K, paired, draws_per_chain):
    """VALU wave-instructions one chain-iteration cannot avoid in this formulation (docs/history.md 6.1): one
    Philox4x32 block per four draws -- per EIGHT for single-end two-isoform events -- (4 x (rounds - 1) = 24 at the contract's 7 rounds, round 0 hoisted), per draw K - 1 compare-and-count pairs
    (single-end; paired-end ~15 per read and compatible isoform for weights, compare, select, score gather), and the
    scalar step's 5K + 3 transcendentals at ~55 instructions each if every lane of a wavefront has one to do, plus
    its ~3K divisions."""
    K = float(K)
    philox = 4.0 * (PHILOX_ROUNDS - 1)    # 2 multiplies + 2 three-input xors per round, round 0 hoisted
    scalar = ((5.0 * K + 3.0) * 55.0 + 3.0 * K * 10.0) / 64.0
    if not paired and K == 2:
        # lazy low bits (include/miso_philox.h): one block per EIGHT draws; per draw a compare-and-count pair on the
        # high half-word and the test for equality with the threshold's
        return draws_per_chain / 512.0 * (philox + 8.0 * 3.0) + scalar
    per_block = philox + (8.0 * (K - 1.0) if not paired else 30.0 * K)
    return draws_per_chain / 256.0 * per_block + scalar


def roofline_for(batch, kernel_ms, key, sh, clock=None):
    """The launch against the bound that applies to it.  The sampler kernels are VALU-issue bound (no HBM
    stream, no MFMA): `achieved` = VALU issue cycles the launch needs per second of kernel time.  It is a MODEL
    (`frac_source`): VALU wave-instructions per chain-iteration and the issue cycles one takes come from the
    committed rocprofv3 PMC passes of this very workload and these very kernels (profiles/valu_model.json,
    tools/prof_summary.py), scaled to this run's chains, iterations and MEASURED kernel time; `peak` = 1024 SIMDs
    x the shader clock THIS run's kernels ran at (`clock`: measure_clock below; 2.4 GHz nominal without it).  The
    model prices a run whose launch took as many shader CYCLES as the profiled one (within 5 %): a box at another
    clock keeps its figure, a changed kernel or launch plan loses it.  A profile of other kernels than the ones this
    run launched is refused (frac null).
    `floor_frac` = the instruction floor / the profiled instructions per chain-iteration: useful issue, not busy
    issue.  Beside them: the share of the Philox ceiling, the measured HBM fraction and SURVEY 8(d)'s
    algorithmic-bytes figure."""
    stats = batch.launch_stats()
    name = batch.last_kernels()
    t = kernel_ms * 1e-3
    alg_bytes = batch.algorithmic_bytes()
    ghz = clock["ghz"] if clock and clock.get("ghz") else None
    peak = SIMDS * ghz if ghz else VALU_PEAK_GCYC
    out = {"bound": "valu", "unit": "Gcycle/s", "peak": round(peak, 1), "peak_nominal": round(VALU_PEAK_GCYC, 1),
           "clock_ghz": None if ghz is None else round(ghz, 4),
           "clock_source": None if not clock else clock.get("source"),
           "kernel": name, "kernel_ms": round(kernel_ms, 3)}
    m = load_json("valu_model.json").get(key)
    chain_iters = sum(k["chains"] * k["iterations"] for k in stats["kernels"])
    chains = sum(k["chains"] for k in stats["kernels"])
    draws = sum(k["words"] for k in stats["kernels"]) / max(chains, 1.0)
    # (rocprofv3 prints every template argument, the library's names leave a defaulted `false` out)
    profiled = None if m is None else sorted(re.search(r"miso::(sampler_[^(]+)\(", k).group(1).replace(", false>", ">")
                                             for k in m["kernels"])
    launched = sorted(set(re.findall(r"sampler_\w+(?:<[^>]*>)?", name)))   # (size buckets may launch one kernel twice)
    # the committed profile's own kernel time (rocprofv3 --kernel-trace of the same command; several kernels side by side:
    # the longest): the model's instructions per chain-iteration only price THIS run if the profiled launch took as long --
    # a kernel changed since, a box at another clock, a different launch plan all show up here
    prof_ms = prof_ghz = None
    if m is not None:
        ns = [v.get("kernel_ns") for v in m["kernels"].values() if v.get("kernel_ns")]
        prof_ms = m["launch_span_ns"] / 1e6 if m.get("launch_span_ns") else (max(ns) / 1e6 if ns else None)
        # the profiled launch's own clock: recorded by the profile pass (tools/prof_summary.py), or -- one kernel alone on
        # the device -- GRBM_GUI_ACTIVE per XCD over the traced duration
        prof_ghz = m.get("clock_ghz")
        if not prof_ghz and len(m["kernels"]) == 1:
            v = next(iter(m["kernels"].values()))
            if v.get("kernel_cycles") and v.get("kernel_ns"):
                prof_ghz = v["kernel_cycles"] / v["kernel_ns"]
    out["profile_kernel_ms"] = None if prof_ms is None else round(prof_ms, 3)
    out["profile_clock_ghz"] = None if not prof_ghz else round(prof_ghz, 4)
    # (several kernels side by side are placed by the dispatcher as it pleases: the same launch scatters +- 15 % from one
    # time to the next -- 758 ... 1056 ms for one kernel of the whole-gene mix over four launches, profiles/r06_pe_mix_summary.txt)
    tol = 0.05 if m is None or len(m["kernels"]) == 1 else 0.20
    if ghz and prof_ghz and prof_ms is not None:     # cycles against cycles
        out["kernel_Mcycles"] = round(kernel_ms * ghz * 1e3, 2)
        out["profile_kernel_Mcycles"] = round(prof_ms * prof_ghz * 1e3, 2)
        stale = abs(prof_ms * prof_ghz - kernel_ms * ghz) > tol * kernel_ms * ghz
        stale_what = "%.2f M shader cycles, this run's %.2f M" % (prof_ms * prof_ghz * 1e3, kernel_ms * ghz * 1e3)
    else:                                            # no clock on one side: milliseconds against milliseconds
        stale = prof_ms is not None and abs(prof_ms - kernel_ms) > tol * kernel_ms
        stale_what = "%.3f ms, this run's %.3f ms" % (prof_ms or 0.0, kernel_ms)
    if m is not None and profiled == launched and not stale:
        cyc = m["valu_per_chain_iteration"] * chain_iters * m["issue_cycles_per_valu"]
        out["achieved"] = round(cyc / t / 1e9, 1)
        out["frac"] = round(cyc / t / 1e9 / peak, 4)
        out["frac_nominal_clock"] = round(cyc / t / 1e9 / VALU_PEAK_GCYC, 4)
        out["frac_source"] = ("model: VALU instructions per chain-iteration x issue cycles per instruction from the committed "
                              "rocprofv3 PMC pass of this workload and these kernels (profiles/valu_model.json <- %s), x this "
                              "run's chain-iterations / this run's measured kernel time" % m["source"])
        out["model"] = {"valu_per_chain_iteration": round(m["valu_per_chain_iteration"], 1),
                        "issue_cycles_per_valu": round(m["issue_cycles_per_valu"], 3),
                        "chain_iterations": chain_iters,
                        "profiled_valu_busy": {k: (None if v["valu_busy"] is None else round(v["valu_busy"], 4))
                                               for k, v in m["kernels"].items()},
                        "profiled_wave_slot_occupancy": {k: (None if v["wave_slot_occupancy"] is None else round(v["wave_slot_occupancy"], 4))
                                                         for k, v in m["kernels"].items()}}
        if not isinstance(sh["K"], (tuple, list)):
            # collapsed Gibbs step: no read sweep; one BTRS trial (sqrt, a logarithm, ~5 divisions, half a Philox block)
            fl = valu_floor(sh["K"], sh["paired"], 0.0) + 150.0 / 64.0 if key.endswith("|collapsed") else valu_floor(sh["K"], sh["paired"], draws)
            out["floor_valu_per_chain_iteration"] = round(fl, 1)
            out["floor_frac"] = round(fl / m["valu_per_chain_iteration"], 4)
    else:
        out["achieved"] = None
        out["frac"] = None
        out["frac_source"] = ("no rocprofv3 PMC pass committed for this workload (%s)" % key) if m is None else \
            ("the committed profile's launch took %s (more than %d %% apart): the model does not price "
             "this run, re-profile" % (stale_what, round(100 * tol))) if (profiled == launched and stale) else \
            ("the committed profile of this workload is of other kernels (%s): re-profile" % ",".join(profiled))
    out["draws_per_chain"] = round(draws, 1)
    # (tools/rng_bench.hip, MI355X: philox4x32-10 2885 G words/s, philox4x32-7 the same generator at 7 / 10 of the rounds)
    rng_ceiling = load_json("rng_ceiling.json").get("philox4x32_7_outputs_per_s", 2885e9 * 10.0 / 7.0)
    # (single-end two-isoform events draw half-words: two reads per generated word, include/miso_philox.h)
    halves = (not sh["paired"]) and sh["K"] == 2 and not key.endswith("|collapsed")
    out["rng_frac"] = round(stats["uniforms"] * (0.5 if halves else 1.0) / t / rng_ceiling, 4)
    out["rng_note"] = "Philox4x32-7 words consumed / s over the chip's ceiling for that generator (tools/rng_bench.hip)"
    traffic = load_json("traffic.json").get(key)
    out["traffic"] = None if traffic is None else traffic["hbm_bytes_per_launch"]
    out["hbm_measured_frac"] = None if traffic is None else round(
        traffic["hbm_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS, 5)
    out["algorithmic_bytes_per_launch"] = alg_bytes
    out["algorithmic_GBs"] = round(alg_bytes / t / 1e9, 1)
    # SURVEY 8(d)'s definition of the fraction, stated for the record: the REFERENCE algorithm's bytes over the HBM peak.
    # Above 1 means the kernel does not move those bytes (the event lives in registers / LDS); `traffic` is what it moves
    out["bytes_frac_8d"] = round(alg_bytes / t / 1e9 / HBM_PEAK_GBS, 2)
    out["algorithmic_note"] = ("SURVEY 8(d) accounting of the reference algorithm ((8K+20)N bytes per "
                               "chain-iteration); the kernels keep the event on chip, so this exceeds the "
                               "HBM peak and is not a utilisation figure")
    return out


# ------------------------------------------------------------------------------------------------
def time_batch(batch, seed, first, steps, warmup, barrier=None):
    def step():
        batch.launch(seed=seed, first_event_id=first)
        return batch.sync()
    for _ in range(warmup):
        step()
    if barrier:
        barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        kernel_ms.append(step())
    if barrier:
        barrier()
    return time.perf_counter() - t0, kernel_ms


def measure_clock(batch, seed, first, kernel_ms):
    """The shader clock the timed launches ran at: ONE more launch of the same batch right behind them, with the library's
    clock probe on (include/miso_amd.h miso_batch_set_clock_probe: one sleeping wavefront beside the sampler kernels that
    reads s_memtime against the constant reference clock over exactly the launch).  Outside the timed region; the probed
    launch's own kernel time is reported so that it can be seen to be one of the timed ones."""
    try:
        batch.set_clock_probe(True)
        batch.launch(seed=seed, first_event_id=first)
        ms = batch.sync()
        ghz, window = batch.last_clock()
    finally:
        batch.set_clock_probe(False)
    # (the probe's wavefront takes registers on one SIMD: a kernel that fills the register file loses a workgroup slot there)
    ok = ghz > 0.0 and abs(ms - kernel_ms) <= (0.06 if "," not in batch.last_kernels() else 0.25) * kernel_ms
    return {"ghz": ghz if ok else None, "probe_kernel_ms": round(ms, 3), "probe_window_ms": round(window, 3),
            "source": ("s_memtime / wall_clock64 of one wavefront beside the kernels of one more launch behind the timed ones "
                       "(%.3f ms; the timed launches' average %.3f ms)" % (ms, kernel_ms)) if ok else
                      ("no clock: probe window %.3f ms, probed launch %.3f ms against %.3f ms timed" % (window, ms, kernel_ms))}


def stream_rows(batch, n_events, sh, first, seed, device):
    """Rows f2 / f3 on the samples where they are (outside the timed region): device-side summaries and the
    two-sample Bayes factors of BASELINE configs[4], with the HBM bytes they stream."""
    S = sh["chains"] * (sh["iters"] - sh["burn"]) // sh["lag"]
    K = sh["K"] if not isinstance(sh["K"], (tuple, list)) else None
    batch.summarize(0.95)                      # first call: code object load
    t1 = time.perf_counter()
    batch.summarize(0.95)
    summary_ms = 1e3 * (time.perf_counter() - t1)
    other = build(first + (1 << 24), n_events, sh, device_match=False)
    other.upload(device)
    other.launch(seed=seed ^ 0x5851F42D4C957F2D, first_event_id=first)
    other.sync()
    batch.compare(other, 0.3)
    t1 = time.perf_counter()
    batch.compare(other, 0.3)
    compare_ms = 1e3 * (time.perf_counter() - t1)
    del other
    out = {"summary_ms": round(summary_ms, 3), "compare_ms": round(compare_ms, 3)}
    if K:
        sb, cb = 8.0 * K * S * n_events, 2 * 8.0 * K * S * n_events
        out["summary_hbm_frac"] = round(sb / (summary_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["compare_hbm_frac"] = round(cb / (compare_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["stream_note"] = ("wall time of the whole call (allocation, launch, result copy, host sync) over the algorithmic sample "
                              "bytes (8 K S per event, twice for the comparison) and 8 TB/s; the kernels alone: DESIGN.md 4.9")
    return out


# Opt-in mode (miso_batch_set_collapsed, csrc/kernels_lane.hip): the single-end two-isoform workloads once more with the
# COLLAPSED Gibbs step -- the count of the exchangeable reads drawn as one exact binomial per iteration instead of one
# uniform per read (same Markov chain on (psi, counts), different draws).  Same workload, same reference runs, same
# |delta psi| test as the row it shadows; "main" = the headline workload.  The headline `value` stays the default mode.
COLLAPSED_ROWS = ("main", "se_k2_hg19", "se_k2_hg19_defaults", "se_k2_defaults")


def matrix_row(a, local_rank, wid, label, sh, n, st, collapsed=False):
    from miso_amd import workload
    spec = reads_spec(sh)
    total_reads = sum(workload.event_n_reads(e, spec) for e in range(n))
    b = build(0, n, sh, collapsed=collapsed)
    b.upload(local_rank)
    # three launches behind one warm-up, each timed on its own, the MEDIAN one reported (kernel time and wall time): a row is a
    # few hundred milliseconds of a four-minute run on a shared pool, and one launch in some hundred is 10 % off for reasons of
    # the box (round 6: `se_k2_defaults` 223 ms once, 199.0 - 200.0 ms in the eleven launches around it)
    time_batch(b, a.seed, 0, 0, 1)
    runs = sorted((time_batch(b, a.seed, 0, 1, 0) for _ in range(3)), key=lambda x: x[0])
    elapsed, kms = runs[1]
    avg = kms[0]
    r = roofline_for(b, avg, workload_key(n, sh) + ("|collapsed" if collapsed else ""), sh, measure_clock(b, a.seed, 0, avg))
    row = {"id": wid, "workload": label, "events": n, "events_per_s": round(n / elapsed, 1),
           "launch_ms_all": [round(1e3 * x[0], 3) for x in runs],
           "reads_iter_per_s": round(1.0 * total_reads * sh["chains"] * sh["iters"] / elapsed, 1),
           "kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "clock_ghz": r["clock_ghz"], "valu_frac": r["frac"],
           "valu_frac_nominal_clock": r.get("frac_nominal_clock"), "floor_frac": r.get("floor_frac"),
           "frac_source": r["frac_source"], "rng_frac": r["rng_frac"], "hbm_measured_frac": r["hbm_measured_frac"],
           "algorithmic_GBs": r["algorithmic_GBs"]}
    if collapsed:
        row["mode"] = "collapsed Gibbs step (opt-in): counts drawn as exact binomials, include/miso_binomial.h"
        row["rng_frac"] = None   # (reads x iterations are not Philox words here)
    if st:
        row["cpu_baseline"] = st["baseline"]
    del b
    if st and len(st["study"]) >= 2 * len(set(r_[0] for r_ in st["study"])):   # at least two reference seeds per event
        row["delta_psi"] = delta_psi_with_retry(wid, sh, st["study"], local_rank, a.seed, collapsed)
    return row


def run_matrix(a, local_rank, studies, main_sh=None):
    rows = []
    only = set(a.matrix_only.split(",")) if a.matrix_only else None
    if main_sh is not None and not main_sh["paired"] and main_sh["K"] == 2 and "main" in COLLAPSED_ROWS and not only and not a.collapsed:
        rows.append(matrix_row(a, local_rank, "main_collapsed", "the headline workload, collapsed Gibbs step", main_sh,
                               min(a.matrix_events, a.events), studies.get("main") if studies else None, collapsed=True))
    for wid, label, ov, events, _ in MATRIX:
        if only and wid not in only and wid + "_collapsed" not in only:
            continue
        sh = dict(BASE_SHAPE, **ov)
        n = min(a.matrix_events, events)
        st = studies.get(wid) if studies else None
        if not only or wid in only:
            rows.append(matrix_row(a, local_rank, wid, label, sh, n, st))
        if wid in COLLAPSED_ROWS and (not only or wid + "_collapsed" in only):
            rows.append(matrix_row(a, local_rank, wid + "_collapsed", label + ", collapsed Gibbs step", sh, n, st, collapsed=True))
    return rows


MATRIX_COLS = ["id", "events_per_s", "kernel_ms", "clock_ghz", "valu_frac", "floor_frac", "hbm_measured_frac", "cpu", "dpsi_pass", "max_z", "p_row"]


def _r(x, n):
    return None if x is None else round(x, n)


def compact_line(full):
    """The one line the driver parses (the contract's keys, `roofline`, `cpu_baseline`, `delta_psi`, one short record per
    matrix row), kept under 4 KB; everything else is in the full record."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: full[k] for k in keep}
    c = full["config"]
    out["config"] = {k: c[k] for k in ("workload", "events_per_gpu", "K", "reads", "iters", "burn_in", "lag", "chains", "collapsed")}
    out["config"]["parallelism"] = "%d cost-balanced contiguous event shards, one process per GPU, no collective" % full["n_gpus"]
    r = full["roofline"] or {}
    kern = r.get("kernel")
    out["roofline"] = {"bound": r.get("bound"), "kernel": None if kern is None else kern[:96], "kernel_ms": r.get("kernel_ms"),
                       "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"),
                       "clock_ghz": r.get("clock_ghz"), "kernel_Mcycles": r.get("kernel_Mcycles"),
                       "profile_kernel_Mcycles": r.get("profile_kernel_Mcycles"),
                       "floor_frac": r.get("floor_frac"), "traffic": _r(r.get("traffic"), 0), "hbm_measured_frac": r.get("hbm_measured_frac"),
                       "algorithmic_GBs": r.get("algorithmic_GBs"), "bytes_frac_8d": r.get("bytes_frac_8d"),
                       "profile_kernel_ms": r.get("profile_kernel_ms"),
                       "frac_source": None if r.get("frac") is None else "model: profiles/valu_model.json x this run's kernel time"}
    b = full.get("cpu_baseline")
    out["cpu_baseline"] = None if b is None else {
        "value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "value_1core": b["value_1core"],
        "sample": b["sample"].split(" (")[0] + ", slowest process" + b["sample"].split("slowest process")[-1][:12]}
    d = full.get("delta_psi")
    out["delta_psi"] = None if d is None else {k: d.get(k) for k in ("pass", "p_row", "p_pooled", "p_sign", "max_z", "n_fail",
                                                                     "n_fail_expected", "mean_abs_dpsi", "max_abs_dpsi", "confirmed_fail")}
    if d is not None:
        out["delta_psi"]["design"] = d["design"].split(";")[0]
    for k in ("per_rank_kernel_ms", "per_rank_elapsed_ms", "summary_ms", "compare_ms", "cpu_reference_wall_s", "stub", "full_record"):
        if full.get(k) is not None:
            out[k] = full[k]
    if full.get("matrix"):
        out["matrix_cols"] = MATRIX_COLS
        rows = []
        for m in full["matrix"]:
            dp = m.get("delta_psi") or {}
            rows.append([m["id"], round(m["events_per_s"]), _r(m["kernel_ms"], 2), _r(m.get("clock_ghz"), 3), _r(m["valu_frac"], 3), _r(m.get("floor_frac"), 3),
                         _r(m["hbm_measured_frac"], 3), _r((m.get("cpu_baseline") or {}).get("value"), 1),
                         "retry_pass" if (dp.get("pass") is False and dp.get("confirmed_fail") is False) else dp.get("pass"),
                         _r(dp.get("max_z"), 2), _r(dp.get("p_row"), 4)])
        out["matrix"] = rows
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--events", type=int, default=40000, help="events per GPU")
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--K-range", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="isoforms per event drawn from [LO, HI] by event id (configs[3] proxy: whole-gene "
                         "mode, mixed isoform counts in one batch); overrides --K")
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--reads-dist", choices=["fixed", "hg19"], default="fixed",
                    help="hg19: heavy-tailed read counts per event (workload.HG19_LIKE) instead of --reads for every event")
    ap.add_argument("--read-len", type=int, default=36)
    ap.add_argument("--iters", type=int, default=7500)
    ap.add_argument("--burn", type=int, default=2500)
    ap.add_argument("--lag", type=int, default=1)
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--collapsed", type=int, default=0, choices=[0, 1, 2],
                    help="single-end, opt-in: the collapsed Gibbs step (miso_batch_set_collapsed): 1 two-isoform events, 2 all")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-events", type=int, default=150,
                    help="reference events per host process (~12 s of CPU work per core at the default shape)")
    ap.add_argument("--dpsi-events", type=int, default=128, help="events of the |delta psi| tolerance study")
    ap.add_argument("--dpsi-seeds", type=int, default=16, help="reference seeds per event of that study")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="no reference runs at all (no cpu_baseline, no delta_psi)")
    ap.add_argument("--no-matrix", action="store_true", help="skip the other shapes (hg19-like, K=5, K=10, defaults, paired-end)")
    ap.add_argument("--matrix-only", default="", help="comma-separated row ids of the matrix to run (default: all)")
    ap.add_argument("--matrix-events", type=int, default=40000)
    ap.add_argument("--no-streams", action="store_true", help="skip the device-side summaries / Bayes factors (rows f2, f3)")
    ap.add_argument("--host-match", action="store_true",
                    help="compute the read x isoform compatibility on the host instead of the GPU (row f1)")
    ap.add_argument("--full-out", default="", help="where the full record goes (default gpurun_out/bench_full.json)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY (tests/test_bench_launch.py): build and shard the events on the host, skip "
                         "every GPU call; the line says \"stub\": true and its value means nothing")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # every rank's native thread pools (event construction, packing) get this rank's share of the host cores, not
    # all of them (N ranks x all cores on N-times-fewer cores each would dominate the run's wall time)
    if world > 1:
        cores = sorted(os.sched_getaffinity(0))
        share = max(1, min(len(cores), usable_cores()) // world)
        mine = cores[(local_rank * share) % len(cores):][:share] or cores[:1]
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            pass

    sh = shape_of(a, K=tuple(a.K_range) if a.K_range else a.K)
    default_shape = not a.K_range and not a.paired and a.reads_dist == "fixed"
    want_matrix = world == 1 and default_shape and not a.no_matrix and not a.stub
    studies = cpu_wall = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.stub:
        wanted = {}
        if default_shape:
            timed = [(e, 42 + e) for e in range(a.cpu_events * usable_cores())]
            study = [(e, 1000003 * (s + 1) + e) for e in range(a.dpsi_events) for s in range(a.dpsi_seeds)] if a.dpsi_seeds > 1 else []
            wanted["main"] = (sh, timed, study)
        else:   # another shape on the command line (profiles): a bounded sample of it, timed on the study's runs
            ne, ns = (32, 1) if a.K_range else (64, 8)
            wanted["main"] = (sh, [], [(e, 1000003 * (s + 1) + e) for e in range(ne) for s in range(ns)])
        if want_matrix:
            only = set(a.matrix_only.split(",")) if a.matrix_only else None
            for wid, _, ov, _, st in MATRIX:
                if st is None or (only and wid not in only):
                    continue
                wanted[wid] = (dict(BASE_SHAPE, **ov), [], [(e, 1000003 * (s + 1) + e) for e in range(st[0]) for s in range(st[1])])
        studies, cpu_wall = cpu_studies(wanted)   # before anything touches the GPU (fork-safe)

    from miso_amd import capi, workload
    dist = None
    if world > 1:
        # control plane only (barrier, max of the elapsed times): CPU tensors over gloo.  The data path
        # has no exchange step, so no RCCL communicator is created.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node: never depend on the hostname resolving
        dist.init_process_group("gloo")
    if not a.stub:
        capi.set_device(local_rank)

    # the rank's shard of the global event list: contiguous, balanced by cost
    n_global = a.events * world
    costs = workload.event_costs(0, n_global, sh["K"], reads_spec(sh), a.iters, a.chains)
    first, last = workload.shard_bounds_by_cost(costs, world, rank)
    n_local = last - first
    t_build = time.perf_counter()
    batch = build(first, n_local, sh, device_match=not a.host_match and not a.stub, collapsed=a.collapsed)
    t_up = time.perf_counter()
    if not a.stub:
        batch.upload(local_rank)   # device_match: read x isoform compatibility on the GPU, then packing
    t_up = time.perf_counter() - t_up
    t_build = time.perf_counter() - t_build

    def barrier():
        if dist is not None:
            dist.barrier()

    if a.stub:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * a.steps)
        barrier()
        elapsed, kernel_ms = time.perf_counter() - t0, [10.0] * a.steps
    else:
        elapsed, kernel_ms = time_batch(batch, a.seed, first, a.steps, a.warmup, barrier)
    shard = [rank, first, last, float(costs[first:last].sum()), sum(kernel_ms) / max(len(kernel_ms), 1), 1e3 * elapsed / max(a.steps, 1)]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        shards = [None] * world
        dist.all_gather_object(shards, shard)
    else:
        shards = [shard]

    rc = 0
    if rank == 0:
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        if a.stub:
            roof = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GCYC, "unit": "Gcycle/s", "frac": None,
                    "traffic": None}
        else:
            roof = roofline_for(batch, avg_ms, workload_key(n_local, sh) + ("|collapsed" if a.collapsed else ""), sh,
                                measure_clock(batch, a.seed, first, avg_ms))
        cpu = delta = None
        if studies and "main" in studies:
            cpu = studies["main"]["baseline"]
            delta = delta_psi_with_retry("main", sh, studies["main"]["study"], local_rank, a.seed, bool(a.collapsed))
            if delta is not None:
                delta["single_runs"] = single_run_deltas(batch, studies["main"]["timed"])
                if delta.get("confirmed_fail"):
                    rc = 3
        streams = {}
        if not a.stub and not a.no_streams and world == 1:
            streams = stream_rows(batch, n_local, sh, first, a.seed, local_rank)
        kind = "paired-end" if a.paired else "skipped-exon single-end"
        rd = "hg19-like read counts (20..1e5)" if a.reads_dist == "hg19" else "%d reads" % a.reads
        if a.K_range:
            wl = ("configs[3] proxy (whole-gene mode): %d %s genes/GPU, %d-%d isoforms" % (a.events, kind, a.K_range[0], a.K_range[1]))
        elif a.paired:
            wl = "configs[2] proxy: %d paired-end events/GPU (insert 250 +- 30), K=%d isoforms" % (a.events, a.K)
        else:
            wl = "configs[1] proxy (hg19 SE set): %d %s events/GPU, K=%d isoforms" % (a.events, kind, a.K)
        wl += ", %s of %d bp, %d iters (%d burn-in + %d kept), lag %d, %d chain(s)" % (
            rd, a.read_len, a.iters, a.burn, a.iters - a.burn, a.lag, a.chains)
        tot = sum(s[3] for s in shards)
        out = {
            "metric": "AS events/sec at 5000 iters (1k reads x 2-10 iso)",
            "value": round(n_global * a.steps / elapsed, 1), "unit": "events/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl, "events_per_gpu": a.events,
                       "K": a.K if not a.K_range else list(a.K_range), "reads": sh["reads"], "iters": a.iters,
                       "burn_in": a.burn, "lag": a.lag, "chains": a.chains, "collapsed": a.collapsed,
                       "parallelism": "%d contiguous cost-balanced event shards, one process per GPU, "
                                      "no collective on the data path" % world,
                       "shards": [[r, lo, hi] for r, lo, hi, *_ in shards],
                       "shard_cost_share": [round(s_[3] / tot, 5) if tot else None for s_ in shards]},
            "roofline": roof,
            "cpu_baseline": cpu,
            "delta_psi": delta,
            # every rank's own kernel time (HIP events on its stream) and wall time of the timed region: a scaling run can
            # tell shard imbalance (kernel_ms differ) from a clock drop under the node's power budget (all grow together)
            "per_rank_kernel_ms": [round(s_[4], 3) for s_ in shards],
            "per_rank_elapsed_ms": [round(s_[5], 3) for s_ in shards],
            "host_build_s": round(t_build, 2), "upload_s": round(t_up, 3),
            "match_kernel_ms": None if a.stub else round(batch.match_ms(), 3),
            "summary_ms": streams.get("summary_ms"), "compare_ms": streams.get("compare_ms"),
            "summary_hbm_frac": streams.get("summary_hbm_frac"), "compare_hbm_frac": streams.get("compare_hbm_frac"),
            "stream_note": streams.get("stream_note"),
            "cpu_reference_wall_s": None if cpu_wall is None else round(cpu_wall, 1),
        }
        if a.stub:
            out["stub"] = True
        elif want_matrix:
            del batch
            out["matrix"] = run_matrix(a, local_rank, studies, sh)
            if any((r.get("delta_psi") or {}).get("confirmed_fail") for r in out["matrix"]):
                rc = 3
        full_path = a.full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
            out["full_record"] = os.path.relpath(full_path, ROOT)
        except OSError as err:
            print("bench.py: full record not written (%s)" % err, file=sys.stderr, flush=True)
        sys.stderr.flush()
        print(json.dumps(compact_line(out), separators=(",", ":")), flush=True)   # the LAST line of stdout
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
