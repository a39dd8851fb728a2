#!/usr/bin/env python3
"""bench.py -- AS events/sec of the MI355X MISO sampler on BASELINE.json's metric shape.

One "step" = one pass of the hot path (all chains, all iterations) over one batch of synthetic
events already resident in HBM.  Headline workload (BASELINE.json configs[1]): skipped-exon events,
2 isoforms, 1000 single-end 36-bp reads, 2500 burn-in + 5000 kept iterations, lag 1, 1 chain.

    python bench.py [--gpus N --steps K --warmup W] [--events E per GPU] [--K 2] [--reads 1000]

`--gpus N` (N > 1) starts N ranks itself -- one fresh process per GPU through
`python -m torch.distributed.run`, before anything in this process touches the GPU (the reference's
dispatcher starts its own workers too, misopy/miso.py:165-187, 205-214) -- unless it is already
running under a launcher (WORLD_SIZE set).  Rank 0 prints ONE JSON line.  The global event list
(N x --events events) is split into N contiguous shards balanced by cost (sum of
chains x iterations x reads), every event keeps its global id, there is no collective on the data
path; ranks only meet in a gloo barrier and a max-over-ranks of the elapsed time.

The line carries, besides the contract's keys:
  roofline            the dominant kernel against the bound that applies to it: VALU issue (VALU
                      instructions per chain-iteration and their issue cycles from the committed
                      rocprofv3 passes, profiles/valu_model.json), with the RNG fraction, the
                      measured HBM fraction and SURVEY 8(d)'s algorithmic-bytes figure beside it
  cpu_baseline        the real reference C core (oracle/_ref) on the host cores, bounded sample
  delta_psi_vs_reference   |delta psi| with a pass/fail: 4 x MCSE from 16 reference seeds per event
  matrix              the other shapes of the metric (K = 5, 10; MISO default settings; paired-end),
                      timed inside the same run
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SIMDS = 256 * 4              # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # max shader clock (MI355X_MICROARCH.md)
VALU_PEAK_GCYC = SIMDS * CLOCK_GHZ   # VALU issue cycles per ns of wall time, whole chip


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


# ------------------------------------------------------------------------------------------------
# CPU side: the real reference (oracle/_ref) or the oracle port, timed and used for |delta psi|
# ------------------------------------------------------------------------------------------------
def _psi_stats(smp):
    """Posterior mean, Chen-Shao 95 % bounds (credible_intervals.py:31-55) and sd of isoform 0."""
    import numpy as np
    x = np.sort(np.asarray(smp)[:, 0])
    n = len(x)
    return (float(x.mean()), float(x[int(round(0.025 * n)) - 1]), float(x[int(round(0.975 * n)) - 1]),
            float(x.std(ddof=1)))


def _cpu_worker(args):
    """One host process: `ev_seeds` = [(event id, reference seed)], run one after the other."""
    kind, ev_seeds, K, n_reads, read_len, iters, burn, lag, chains = args
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)  # the reference prints "no chains: %d" per call (miso.c:837)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import OrcLib, RefLib
    from miso_amd import workload
    L = RefLib() if kind == "reference" else OrcLib()
    probs = {}
    for e, _ in ev_seeds:
        if e not in probs:
            exons, isoforms, pos, cig = workload.event_reads(e, K, n_reads, read_len)
            probs[e] = (L.gene([c for ex in exons for c in ex], isoforms), pos, cig)
    out = []
    busy = 0.0
    for e, seed in ev_seeds:
        g, pos, cig = probs[e]
        L.rng_seed(seed)   # the reference's generator is one global stream (random.c:491)
        t0 = time.perf_counter()
        r = L.miso(g, pos, cig, read_len, iters=iters, burn=burn, lag=lag, chains=chains)
        busy += time.perf_counter() - t0
        assert r.rc == 0
        out.append((e, seed) + _psi_stats(r.samples))
    return busy, out


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


def cpu_reference(a):
    """(cpu_baseline dict, timing-sample rows, seed-study rows).  P processes (the reference's own
    parallelism is processes, misopy/miso.py:165-187).  Phase 1, timed: --cpu-events events per
    process under one seed.  Phase 2, not timed: the first --dpsi-events events under --dpsi-seeds
    further seeds each, from which the Monte-Carlo standard error of one run's posterior mean is
    estimated event by event (tolerance 4 x MCSE from >= 8 reference seeds)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import RefLib
    kind = "reference" if RefLib.available() else "port"
    cores = usable_cores()
    shape = (a.K, a.reads, a.read_len, a.iters, a.burn, a.lag, a.chains)
    per_proc = a.cpu_events
    jobs = [(kind, [(e, 42 + e) for e in range(p * per_proc, (p + 1) * per_proc)]) + shape for p in range(cores)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        times = [r[0] for r in res]
        study = []
        if a.dpsi_events > 0 and a.dpsi_seeds > 1:
            pairs = [(e, 1000003 * (s + 1) + e) for e in range(a.dpsi_events) for s in range(a.dpsi_seeds)]
            jobs2 = [(kind, pairs[p::cores]) + shape for p in range(cores)]
            t1 = time.perf_counter()
            study = [row for r in pool.map(_cpu_worker, jobs2) for row in r[1]]
            study_s = time.perf_counter() - t1
        else:
            study_s = 0.0
    busy = max(times)
    base = {"value": round(cores * per_proc / busy, 3), "unit": "events/s", "cores": cores, "kind": kind,
            "sample": "%d events/process x %d processes (same synthetic events, K=%d N=%d iters=%d burn=%d "
                      "lag=%d chains=%d), slowest process %.2fs, pool wall %.2fs; + %d events x %d seeds for "
                      "the |dpsi| tolerance (%.1fs, not timed)"
                      % (per_proc, cores, a.K, a.reads, a.iters, a.burn, a.lag, a.chains, busy, wall,
                         a.dpsi_events, a.dpsi_seeds, study_s),
            "value_1core": round(per_proc / (sum(times) / len(times)), 3)}
    return base, [row for r in res for row in r[1]], study


def delta_psi(batch, a, sample_rows, study_rows):
    """BASELINE's "|delta psi| vs ref" with a pass/fail.  GPU = device-side posterior mean and
    Chen-Shao bounds of isoform 0 (summarize_kernel); reference = the real C core on the same events
    under independent random streams."""
    import numpy as np
    batch.summarize(0.95)
    n_local = len(batch)
    rows = [r for r in sample_rows if r[0] < n_local]
    d_mean, d_lo, d_hi = [], [], []
    for e, _, m, lo, hi, sd in rows:
        gm, glo, ghi = batch.summary(e)
        d_mean.append(abs(gm[0] - m)); d_lo.append(abs(glo[0] - lo)); d_hi.append(abs(ghi[0] - hi))
    out = {"events": len(rows), "mean_abs_dpsi": round(float(np.mean(d_mean)), 6),
           "max_abs_dpsi": round(float(np.max(d_mean)), 6),
           "mean_abs_dci_low": round(float(np.mean(d_lo)), 6),
           "mean_abs_dci_high": round(float(np.mean(d_hi)), 6)}
    by_event = {}
    for e, _, m, lo, hi, sd in study_rows:
        if e < n_local:
            by_event.setdefault(e, []).append((m, sd))
    if by_event:
        S = a.dpsi_seeds
        z, worst = [], None
        for e, runs in sorted(by_event.items()):
            means = np.array([r[0] for r in runs])
            mcse = float(means.std(ddof=1))            # sd of one run's posterior mean over seeds
            post_sd = float(np.mean([r[1] for r in runs]))
            gm = batch.summary(e)[0][0]
            # GPU run (one chain, same MCSE) minus the mean of S reference runs
            tol = 4.0 * mcse * math.sqrt(1.0 + 1.0 / len(means))
            d = abs(gm - float(means.mean()))
            rec = {"event": e, "abs_dpsi": round(d, 6), "tolerance": round(tol, 6), "mcse": round(mcse, 6),
                   "posterior_sd": round(post_sd, 6), "ref_seed_range": round(float(np.ptp(means)), 6)}
            z.append(d / max(tol / 4.0, 1e-300))
            if worst is None or d / max(tol, 1e-300) > worst["abs_dpsi"] / max(worst["tolerance"], 1e-300):
                worst = rec
        z = np.array(z)
        n_fail = int((z > 4.0).sum())
        # MCSE comes from S seeds, so the statistic is Student-t with S-1 degrees of freedom, not
        # normal: P(|t_15| > 4) = 1.2e-3 per event.  The run fails when more events exceed 4 x MCSE
        # than that explains (binomial tail < 1e-3) or when any event is beyond 8 x MCSE.
        from scipy import stats
        p1 = 2 * stats.t.sf(4.0, S - 1)
        allowed = int(stats.binom.isf(1e-3, len(z), p1))
        out.update({"tolerance": "4*MCSE", "mcse_from": "%d reference seeds per event, %d events" % (S, len(z)),
                    "n_fail": n_fail, "n_fail_allowed": allowed,
                    "expected_exceedances_t%d" % (S - 1): round(len(z) * p1, 3),
                    "max_z_in_mcse": round(float(z.max()), 3), "worst_event": worst,
                    "pass": bool(n_fail <= allowed and float(z.max()) <= 8.0)})
    out["note"] = ("isoform 0; independent random streams, so differences are Monte-Carlo error: the K=2 "
                   "proposal step is 0.05 in logit space (miso.c:188, 328), so events with few informative "
                   "reads mix slowly and two runs of the REFERENCE differ by the same amounts "
                   "(worst_event.ref_seed_range)")
    return out


# ------------------------------------------------------------------------------------------------
# roofline of one timed batch
# ------------------------------------------------------------------------------------------------
def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}


def roofline_for(batch, kernel_ms, workload_key):
    """The launch against the bound that applies to it.  The sampler kernels are VALU-issue bound (no HBM
    stream, no MFMA): `achieved` = VALU issue cycles the launch needs per second of kernel time, priced
    from the committed rocprofv3 passes of this very workload (profiles/valu_model.json: VALU
    wave-instructions per chain-iteration x the issue cycles one takes, tools/prof_summary.py) and scaled
    to this run's chains, iterations and measured kernel time; `peak` = 1024 SIMDs x 2.4 GHz.  Beside it:
    the share of the Philox ceiling, the measured HBM fraction and SURVEY 8(d)'s algorithmic-bytes figure."""
    stats = batch.launch_stats()
    name = batch.last_kernels()
    t = kernel_ms * 1e-3
    alg_bytes = batch.algorithmic_bytes()
    out = {"bound": "valu", "unit": "Gcycle/s", "peak": round(VALU_PEAK_GCYC, 1),
           "kernel": name, "kernel_ms": round(kernel_ms, 3)}
    m = load_json("valu_model.json").get(workload_key)
    chain_iters = sum(k["chains"] * k["iterations"] for k in stats["kernels"])
    if m is not None:
        cyc = m["valu_per_chain_iteration"] * chain_iters * m["issue_cycles_per_valu"]
        out["achieved"] = round(cyc / t / 1e9, 1)
        out["frac"] = round(cyc / t / 1e9 / VALU_PEAK_GCYC, 4)
        out["model"] = {"valu_per_chain_iteration": round(m["valu_per_chain_iteration"], 1),
                        "issue_cycles_per_valu": round(m["issue_cycles_per_valu"], 3),
                        "chain_iterations": chain_iters, "source": "profiles/valu_model.json <- " + m["source"],
                        "profiled_valu_busy": {k: (None if v["valu_busy"] is None else round(v["valu_busy"], 4))
                                               for k, v in m["kernels"].items()},
                        "profiled_wave_slot_occupancy": {k: (None if v["wave_slot_occupancy"] is None else round(v["wave_slot_occupancy"], 4))
                                                         for k, v in m["kernels"].items()}}
    else:
        out["achieved"] = None
        out["frac"] = None
        out["model"] = "no rocprofv3 PMC pass committed for this workload (%s)" % workload_key
    rng_ceiling = load_json("rng_ceiling.json").get("philox4x32_10_outputs_per_s", 2885e9)
    out["rng_frac"] = round(stats["uniforms"] / t / rng_ceiling, 4)
    out["rng_note"] = "Philox4x32-10 words consumed / s over the chip's measured ceiling (tools/rng_bench.hip)"
    traffic = load_json("traffic.json").get(workload_key)
    out["traffic"] = None if traffic is None else traffic["hbm_bytes_per_launch"]
    out["hbm_measured_frac"] = None if traffic is None else round(
        traffic["hbm_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS, 5)
    out["algorithmic_bytes_per_launch"] = alg_bytes
    out["algorithmic_GBs"] = round(alg_bytes / t / 1e9, 1)
    out["algorithmic_note"] = ("SURVEY 8(d) accounting of the reference algorithm ((8K+20)N bytes per "
                               "chain-iteration); the kernels keep the event on chip, so this exceeds the "
                               "HBM peak and is not a utilisation figure")
    return out


def traffic_key(kernel, events, K, reads, iters, chains, paired):
    """One canonical key per workload: K as "lo-hi" for isoform ranges, reads as a number or "hg19"."""
    k = "%d-%d" % tuple(K) if isinstance(K, (tuple, list)) else str(K)
    return "%s|events=%d|K=%s|reads=%s|iters=%d|chains=%d|paired=%d" % (
        kernel, events, k, reads, iters, chains, int(paired))


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


MATRIX = [  # (label, overrides): the other shapes BASELINE's metric names, 40 000 events each
    # read counts as a real annotation sees them (workload.HG19_LIKE: log-normal, median 300, 20 ... 10^5 reads per
    # event, a handful of 10^4 ... 10^5-read events per 40 000): "reads_iter_per_s" is what to compare with the uniform rows
    ("SE K=2, hg19-like read counts (20..1e5 per event), 1 chain, 7500 iters", dict(K=2, reads="hg19")),
    ("SE K=2, hg19-like read counts, MISO defaults (6 chains, 5000 iters, 500 burn-in, lag 10)",
     dict(K=2, reads="hg19", chains=6, iters=5000, burn=500, lag=10)),
    ("PE K=2, hg19-like read counts, 1 chain, 7500 iters", dict(K=2, paired=True, reads="hg19")),
    ("SE K=5, hg19-like read counts, 1 chain, 7500 iters", dict(K=5, reads="hg19")),
    ("SE K=5, 1 chain, 7500 iters", dict(K=5)),
    ("SE K=10, 1 chain, 7500 iters", dict(K=10)),
    ("SE K=2, MISO defaults (6 chains, 5000 iters, 500 burn-in, lag 10)", dict(K=2, chains=6, iters=5000, burn=500, lag=10)),
    ("PE K=2 (mean 250, sd 30), 1 chain, 7500 iters", dict(K=2, paired=True)),
    # BASELINE configs[3]: whole-gene mode, 3-20 isoforms per gene, paired-end; "events" are genes here
    ("PE K=3..20 per gene (whole-gene mix), 1 chain, 7500 iters", dict(K=(3, 20), paired=True, events=16384)),
]


def run_matrix(a, local_rank):
    from miso_amd import workload
    rows = []
    for label, ov in MATRIX:
        cfg = dict(K=a.K, reads=a.reads, read_len=a.read_len, iters=a.iters, burn=a.burn, lag=a.lag,
                   chains=a.chains, paired=False)
        cfg.update(ov)
        n = min(a.matrix_events, cfg.get("events", a.matrix_events))
        reads_spec = workload.HG19_LIKE if cfg["reads"] == "hg19" else cfg["reads"]
        total_reads = sum(workload.event_n_reads(e, reads_spec) for e in range(n))
        b = workload.build_batch(0, n, K=cfg["K"], n_reads=reads_spec, read_len=cfg["read_len"],
                                 iters=cfg["iters"], burn=cfg["burn"], lag=cfg["lag"], chains=cfg["chains"],
                                 paired=cfg["paired"], device_match=True)
        b.upload(local_rank)
        elapsed, kms = time_batch(b, a.seed, 0, 2, 1)
        avg = sum(kms) / len(kms)
        r = roofline_for(b, avg, traffic_key(b.last_kernels(), n, cfg["K"], cfg["reads"], cfg["iters"],
                                             cfg["chains"], cfg["paired"]))
        rows.append({"workload": label, "events": n, "events_per_s": round(2 * n / elapsed, 1),
                     "reads_iter_per_s": round(2.0 * total_reads * cfg["chains"] * cfg["iters"] / elapsed, 1),
                     "kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "valu_frac": r["frac"],
                     "rng_frac": r["rng_frac"], "hbm_measured_frac": r["hbm_measured_frac"],
                     "algorithmic_GBs": r["algorithmic_GBs"]})
        del b
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--events", type=int, default=40000, help="events per GPU")
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--K-range", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="isoforms per event drawn from [LO, HI] by event id (configs[3] proxy: whole-gene "
                         "mode, mixed isoform counts in one batch); overrides --K, no CPU baseline")
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--reads-dist", choices=["fixed", "hg19"], default="fixed",
                    help="hg19: heavy-tailed read counts per event (workload.HG19_LIKE) instead of --reads for every event")
    ap.add_argument("--read-len", type=int, default=36)
    ap.add_argument("--iters", type=int, default=7500)
    ap.add_argument("--burn", type=int, default=2500)
    ap.add_argument("--lag", type=int, default=1)
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-events", type=int, default=150,
                    help="reference events per host process (~12 s of CPU work per core at the default shape)")
    ap.add_argument("--dpsi-events", type=int, default=128, help="events of the |delta psi| tolerance study")
    ap.add_argument("--dpsi-seeds", type=int, default=16, help="reference seeds per event of that study")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matrix", action="store_true", help="skip the other shapes (K=5, K=10, defaults, paired-end)")
    ap.add_argument("--matrix-events", type=int, default=40000)
    ap.add_argument("--host-match", action="store_true",
                    help="compute the read x isoform compatibility on the host instead of the GPU (row f1)")
    ap.add_argument("--compare", action="store_true",
                    help="also sample a second RNA-seq sample of the same events and time the device-side "
                         "Bayes factors (BASELINE configs[4]; outside the timed region)")
    ap.add_argument("--summarize", action="store_true",
                    help="also time the device-side posterior summaries (outside the timed region)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY (tests/test_bench_launch.py): build and shard the events on the host, skip "
                         "every GPU call; the line says \"stub\": true and its value means nothing")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    default_shape = not a.K_range and not a.paired and a.reads_dist == "fixed"
    if a.K_range:
        a.no_cpu_baseline = True
    k_spec = tuple(a.K_range) if a.K_range else a.K
    cpu = sample_rows = study_rows = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and default_shape and not a.stub:
        cpu, sample_rows, study_rows = cpu_reference(a)  # before anything touches the GPU (fork-safe)

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
    reads_spec = workload.HG19_LIKE if a.reads_dist == "hg19" else a.reads
    costs = workload.event_costs(0, n_global, k_spec, reads_spec, a.iters, a.chains)
    first, last = workload.shard_bounds_by_cost(costs, world, rank)
    n_local = last - first
    t_build = time.perf_counter()
    batch = workload.build_batch(first, n_local, K=k_spec, n_reads=reads_spec, read_len=a.read_len,
                                 iters=a.iters, burn=a.burn, lag=a.lag, chains=a.chains,
                                 paired=a.paired, device_match=not a.host_match and not a.stub)
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
    shard = [rank, first, last]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        shards = [None] * world
        dist.all_gather_object(shards, shard)
    else:
        shards = [shard]

    summary_ms = compare_ms = None
    if a.summarize and not a.stub:
        t1 = time.perf_counter()
        batch.summarize(0.95)
        summary_ms = 1e3 * (time.perf_counter() - t1)
    if a.compare and not a.stub:
        other = workload.build_batch(first + (1 << 24), n_local, K=a.K, n_reads=reads_spec,
                                     read_len=a.read_len, iters=a.iters, burn=a.burn, lag=a.lag,
                                     chains=a.chains, paired=a.paired)
        other.upload(local_rank)
        other.launch(seed=a.seed ^ 0x5851F42D4C957F2D, first_event_id=first)
        other.sync()
        t1 = time.perf_counter()
        batch.compare(other, 0.3)
        compare_ms = 1e3 * (time.perf_counter() - t1)
        del other

    rc = 0
    if rank == 0:
        delta = None
        if cpu is not None and sample_rows:
            delta = delta_psi(batch, a, sample_rows, study_rows)
            if delta.get("pass") is False:
                rc = 3
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        if a.stub:
            roof = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GCYC, "unit": "Gcycle/s", "frac": None,
                    "traffic": None}
        else:
            roof = roofline_for(batch, avg_ms, traffic_key(batch.last_kernels(), n_local, k_spec,
                                                           "hg19" if a.reads_dist == "hg19" else a.reads, a.iters, a.chains, a.paired))
        kind = "paired-end" if a.paired else "skipped-exon single-end"
        if a.K_range:
            wl = ("configs[3] proxy (whole-gene mode): %d %s genes/GPU, %d-%d isoforms" % (a.events, kind, a.K_range[0], a.K_range[1]))
        elif a.paired:
            wl = "configs[2] proxy: %d paired-end events/GPU (insert 250 +- 30), K=%d isoforms" % (a.events, a.K)
        else:
            wl = "configs[1] proxy (hg19 SE set): %d %s events/GPU, K=%d isoforms" % (a.events, kind, a.K)
        wl += ", %d reads of %d bp, %d iters (%d burn-in + %d kept), lag %d, %d chain(s)" % (
            a.reads, a.read_len, a.iters, a.burn, a.iters - a.burn, a.lag, a.chains)
        out = {
            "metric": "AS events/sec at 5000 iters (1k reads x 2-10 iso)",
            "value": round(n_global * a.steps / elapsed, 1), "unit": "events/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl, "events_per_gpu": a.events,
                       "K": a.K if not a.K_range else list(a.K_range), "reads": a.reads, "iters": a.iters,
                       "burn_in": a.burn, "lag": a.lag, "chains": a.chains,
                       "parallelism": "%d contiguous cost-balanced event shards, one process per GPU, "
                                      "no collective on the data path" % world,
                       "shards": [[r, lo, hi] for r, lo, hi in shards]},
            "roofline": roof,
            "cpu_baseline": cpu,
            "delta_psi_vs_reference": delta,
            "host_build_s": round(t_build, 2), "upload_s": round(t_up, 3),
            "match_kernel_ms": None if a.stub else round(batch.match_ms(), 3),
            "summary_ms": None if summary_ms is None else round(summary_ms, 3),
            "compare_ms": None if compare_ms is None else round(compare_ms, 3),
        }
        if a.stub:
            out["stub"] = True
        elif world == 1 and default_shape and not a.no_matrix:
            del batch
            out["matrix"] = run_matrix(a, local_rank)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
