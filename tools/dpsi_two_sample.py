"""CPU study (this container: needs oracle/_ref): the device contract -- the checker's counter (or collapsed) mode, which
the GPU kernels equal bit for bit -- against the REAL reference, both under several independent random streams per event,
compared with tests/_dpsi.py's permutation tests.  The same test runs in bench.py with the GPU in the checker's place.

    python tools/dpsi_two_sample.py [--K 5] [--reads 1000|hg19] [--paired] [--events 64] [--seeds 8] [--mode counter|collapsed|stream]
                                    [--chains 1 --iters 7500 --burn 2500 --lag 1]
`--mode stream` = the reference against itself (the checker's stream mode IS the reference bit for bit): the null.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    kind, sh, ev_seeds = job
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    import numpy as np
    from _libs import OrcLib, RefLib
    from miso_amd import workload
    import bench
    L = RefLib() if kind == "reference" else OrcLib()
    out = []
    kw = dict(iters=sh["iters"], burn=sh["burn"], lag=sh["lag"], chains=sh["chains"], stop=sh.get("stop", 0),
              max_iters=sh.get("max_iters", 100000))
    if not sh["paired"]:
        kw["algo"] = sh.get("algo", 0)
    cache = {}
    for e, seed in ev_seeds:
        if e not in cache:
            exons, isoforms, pos, cig = workload.event_reads(e, sh["K"], bench.reads_spec(sh), sh["read_len"], sh["paired"],
                                                             sh["mean"], sh["var"])
            cache[e] = (L.gene([c for ex in exons for c in ex], isoforms), pos, cig)
        g, pos, cig = cache[e]
        if kind == "reference" or kind == "stream":
            L.rng_seed(seed)
            extra = {}
        else:
            extra = dict(mode=OrcLib.COUNTER if kind == "counter" else OrcLib.COLLAPSED, seed=seed, event_id=e)
        if sh["paired"]:
            r = L.miso_paired(g, pos, cig, sh["read_len"], sh["mean"], sh["var"], **kw, **extra)
        else:
            r = L.miso(g, pos, cig, sh["read_len"], **kw, **extra)
        assert r.rc == 0
        out.append((e, seed) + tuple(bench._psi_stats(r.samples)[:3]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--K-range", type=int, nargs=2, default=None)
    ap.add_argument("--reads", default="1000")
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--events", type=int, default=64)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--mode", default="counter", choices=["counter", "collapsed", "stream"])
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--iters", type=int, default=7500)
    ap.add_argument("--burn", type=int, default=2500)
    ap.add_argument("--lag", type=int, default=1)
    ap.add_argument("--perm", type=int, default=9999)
    ap.add_argument("--algo", type=int, default=0, help="1 = MARGINAL (single-end)")
    ap.add_argument("--stop", type=int, default=0, help="1 = CONVERGENT_MEAN")
    ap.add_argument("--max-iters", type=int, default=100000)
    a = ap.parse_args()
    import bench
    import _dpsi
    sh = dict(bench.BASE_SHAPE, K=tuple(a.K_range) if a.K_range else a.K, reads=("hg19" if a.reads == "hg19" else int(a.reads)),
              paired=a.paired, chains=a.chains, iters=a.iters, burn=a.burn, lag=a.lag, algo=a.algo, stop=a.stop,
              max_iters=a.max_iters)
    cores = bench.usable_cores()
    ref = [(e, 1000003 * (s + 1) + e) for e in range(a.events) for s in range(a.seeds)]
    oth = [(e, 7000003 * (s + 1) + e) for e in range(a.events) for s in range(a.seeds)]
    jobs = [("reference", sh, ref[p::cores]) for p in range(cores)] + [(a.mode, sh, oth[p::cores]) for p in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(work, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    runs_ref, runs_oth = {}, {}
    for j, part in zip(jobs, res):
        dst = runs_ref if j[0] == "reference" else runs_oth
        for e, _, m, lo, hi in part:
            dst.setdefault(e, []).append((m, lo, hi))
    events = sorted(runs_ref)
    kmax = max(len(runs_ref[e][0][0]) for e in events)
    r = _dpsi.two_sample(_dpsi.stack_runs(runs_oth, events, kmax), _dpsi.stack_runs(runs_ref, events, kmax), n_perm=a.perm)
    r["shape"] = {k: sh[k] for k in ("K", "reads", "paired", "chains", "iters", "burn", "lag", "algo", "stop", "max_iters")}
    r["mode"] = a.mode
    r["cpu_wall_s"] = round(wall, 1)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
