#!/usr/bin/env python3
"""bench.py -- AS events/sec of the MI355X MISO sampler on BASELINE.json's metric shape.

One "step" = one pass of the hot path (all chains, all iterations) over one batch of synthetic
events already resident in HBM.  Workload (BASELINE.json configs[1]): skipped-exon events,
2 isoforms, 1000 single-end 36-bp reads, 2500 burn-in + 5000 kept iterations, lag 1, 1 chain.

    python bench.py [--gpus N --steps K --warmup W] [--events E per GPU] [--K 2] [--reads 1000]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Events are sharded statically over ranks (weak scaling: every GPU
gets --events events with global ids rank*E ..); there is no collective on the data path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _cpu_worker(args):
    """Time the REAL reference (oracle/_ref) or the oracle port on a few events, one process."""
    kind, ev_ids, K, n_reads, read_len, iters, burn, lag, chains = args
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)  # the reference prints "no chains: %d" per call (miso.c:837)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import OrcLib, RefLib
    from miso_amd import workload
    L = RefLib() if kind == "reference" else OrcLib()
    L.rng_seed(42)
    probs = []
    for e in ev_ids:
        exons, isoforms, pos, cig = workload.event_reads(e, K, n_reads, read_len)
        g = L.gene([c for ex in exons for c in ex], isoforms)
        probs.append((g, pos, cig))
    import numpy as np
    stats = []
    t0 = time.perf_counter()
    for g, pos, cig in probs:
        r = L.miso(g, pos, cig, read_len, iters=iters, burn=burn, lag=lag, chains=chains)
        assert r.rc == 0
        stats.append(r.samples)
    dt = time.perf_counter() - t0
    # posterior mean and Chen-Shao 95 % bounds of isoform 0 (credible_intervals.py:31-55), outside
    # the timed region: what the GPU's numbers are compared with (|delta psi| in BASELINE's metric)
    out = []
    for e, smp in zip(ev_ids, stats):
        x = np.sort(np.asarray(smp)[:, 0])
        n = len(x)
        out.append((e, float(x.mean()), float(x[int(round(0.025 * n)) - 1]),
                    float(x[int(round(0.975 * n)) - 1]), float(x.std(ddof=1))))
    return dt, out


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


def cpu_baseline(a):
    """Reference C sampler on the host cores: P processes (the reference's own parallelism is
    processes, misopy/miso.py:165-187), a bounded sample of the bench's own events."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import RefLib
    kind = "reference" if RefLib.available() else "port"
    cores = usable_cores()
    per_proc = a.cpu_events
    jobs = [(kind, list(range(p * per_proc, (p + 1) * per_proc)), a.K, a.reads, a.read_len,
             a.iters, a.burn, a.lag, a.chains) for p in range(cores)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    times = [r[0] for r in res]
    cpu_baseline.psi = [row for r in res for row in r[1]]   # (event, mean, lo, hi, sd) of isoform 0
    busy = max(times)
    return {"value": round(cores * per_proc / busy, 3), "unit": "events/s", "cores": cores,
            "kind": kind,
            "sample": "%d events/process x %d processes (same synthetic events, K=%d N=%d "
                      "iters=%d burn=%d lag=%d chains=%d), slowest process %.2fs, pool wall %.2fs"
                      % (per_proc, cores, a.K, a.reads, a.iters, a.burn, a.lag, a.chains, busy, wall),
            "value_1core": round(per_proc / (sum(times) / len(times)), 3)}


def measured_traffic(kernel, a):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/prof_summary.py from separate FETCH_SIZE / WRITE_SIZE
    runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md).  None when no profile
    of exactly this workload is committed."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    key = "%s|events=%d|K=%d|reads=%d|iters=%d|chains=%d|paired=%d" % (
        kernel, a.events, a.K, a.reads, a.iters, a.chains, int(a.paired))
    rec = table.get(key)
    return None if rec is None else rec["hbm_bytes_per_launch"]


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
    ap.add_argument("--read-len", type=int, default=36)
    ap.add_argument("--iters", type=int, default=7500)
    ap.add_argument("--burn", type=int, default=2500)
    ap.add_argument("--lag", type=int, default=1)
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-events", type=int, default=150,
                    help="reference events per host process (~12 s of CPU work per core at the default shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-match", action="store_true",
                    help="compute the read x isoform compatibility on the host instead of the GPU (row f1)")
    ap.add_argument("--compare", action="store_true",
                    help="also sample a second RNA-seq sample of the same events and time the device-side "
                         "Bayes factors (BASELINE configs[4]; outside the timed region)")
    ap.add_argument("--summarize", action="store_true",
                    help="also time the device-side posterior summaries (outside the timed region)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if a.K_range:
        a.no_cpu_baseline = True
    k_spec = tuple(a.K_range) if a.K_range else a.K
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.paired:
        cpu = cpu_baseline(a)  # before anything touches the GPU (fork-safe)

    from miso_amd import capi, workload
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # launched by torchrun
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    capi.set_device(local_rank)

    first = rank * a.events
    t_build = time.perf_counter()
    batch = workload.build_batch(first, a.events, K=k_spec, n_reads=a.reads, read_len=a.read_len,
                                 iters=a.iters, burn=a.burn, lag=a.lag, chains=a.chains,
                                 paired=a.paired, device_match=not a.host_match)
    t_up = time.perf_counter()
    batch.upload(local_rank)       # device_match: read x isoform compatibility on the GPU, then packing
    t_up = time.perf_counter() - t_up
    t_build = time.perf_counter() - t_build

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    def step():
        batch.launch(seed=a.seed, first_event_id=first)
        return batch.sync()

    for _ in range(a.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        kernel_ms.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    summary_ms = None
    if a.summarize:
        t1 = time.perf_counter()
        batch.summarize(0.95)
        summary_ms = 1e3 * (time.perf_counter() - t1)

    compare_ms = None
    if a.compare:
        other = workload.build_batch(first + (1 << 24), a.events, K=a.K, n_reads=a.reads,
                                     read_len=a.read_len, iters=a.iters, burn=a.burn, lag=a.lag,
                                     chains=a.chains, paired=a.paired)
        other.upload(local_rank)
        other.launch(seed=a.seed ^ 0x5851F42D4C957F2D, first_event_id=first)
        other.sync()
        t1 = time.perf_counter()
        batch.compare(other, 0.3)
        compare_ms = 1e3 * (time.perf_counter() - t1)

    delta = None
    if rank == 0 and cpu is not None and getattr(cpu_baseline, "psi", None):
        # BASELINE's "|delta psi| vs ref": the GPU's posterior summaries (computed on the device) of
        # the very events the reference just sampled on the host, different random streams
        import numpy as np
        batch.summarize(0.95)
        rows = [r for r in cpu_baseline.psi if r[0] < a.events]
        d_mean, d_lo, d_hi, z = [], [], [], []
        for e, m, lo, hi, sd in rows:
            gm, glo, ghi = batch.summary(e)
            d_mean.append(abs(gm[0] - m)); d_lo.append(abs(glo[0] - lo)); d_hi.append(abs(ghi[0] - hi))
            z.append(abs(gm[0] - m) / max(sd, 1e-12))
        delta = {"events": len(rows), "mean_abs_dpsi": round(float(np.mean(d_mean)), 6),
                 "max_abs_dpsi": round(float(np.max(d_mean)), 6),
                 "mean_abs_dci_low": round(float(np.mean(d_lo)), 6),
                 "mean_abs_dci_high": round(float(np.mean(d_hi)), 6),
                 "max_dpsi_in_posterior_sd": round(float(np.max(z)), 4),
                 "note": "isoform 0, GPU (device-side mean / Chen-Shao 95%% bounds) vs the real "
                         "reference C core on the same events; independent random streams, so the "
                         "difference is Monte-Carlo error of two %d-sample chains" % (a.iters - a.burn)}
    if rank == 0:
        total_events = a.events * world * a.steps
        value = total_events / elapsed
        alg_bytes = batch.algorithmic_bytes()          # per launch, this rank
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "AS events/sec at 5000 iters (1k reads x 2-10 iso)",
            "value": round(value, 1), "unit": "events/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1] proxy: %d %s events/GPU, K=%d isoforms, %d reads of "
                                   "%d bp, %d iters (%d burn-in + %d kept), lag %d, %d chain(s)"
                                   % (a.events, "paired-end" if a.paired else "skipped-exon single-end",
                                      a.K if not a.K_range else -1, a.reads, a.read_len, a.iters, a.burn, a.iters - a.burn,
                                      a.lag, a.chains),
                       "events_per_gpu": a.events, "K": a.K if not a.K_range else list(a.K_range), "reads": a.reads, "iters": a.iters,
                       "burn_in": a.burn, "lag": a.lag, "chains": a.chains,
                       "parallelism": "static event shard x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(batch.last_kernels(), a),
                         "kernel": batch.last_kernels(), "kernel_ms": round(avg_ms, 3),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = what the reference algorithm streams "
                                 "(SURVEY 8d: (8K+20)N per chain-iteration); the kernel keeps the "
                                 "event on chip, so frac can exceed 1 and is NOT an HBM-utilisation "
                                 "claim -- see DESIGN.md"},
            "cpu_baseline": cpu,
            "delta_psi_vs_reference": delta,
            "host_build_s": round(t_build, 2), "upload_s": round(t_up, 3),
            "match_kernel_ms": round(batch.match_ms(), 3),
            "summary_ms": None if summary_ms is None else round(summary_ms, 3),
            "compare_ms": None if compare_ms is None else round(compare_ms, 3),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
