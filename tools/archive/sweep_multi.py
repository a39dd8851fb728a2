"""GPU experiment (round 3): sampler_k2_multi (a lane width per event) against the single-/two-width launches on the
uniform and the hg19-like read-count workloads, and the calibration of the planner's cost model
(plan.hpp: VALU per wavefront step and per Philox block) from forced single-width launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import capi, workload

E = int(os.environ.get("SWEEP_EVENTS", 40000))
PAIRED = bool(int(os.environ.get("SWEEP_PAIRED", "0")))
what = (sys.argv[1:] or ["ab", "calib"]) if __name__ == "__main__" else []


def run(b, label, **env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        b.launch(seed=42); b.sync()
        b.launch(seed=42); ms = b.sync()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    st = b.launch_stats()["kernels"]
    print("%-46s %-28s %9.2f ms  %9.0f events/s  waves %d" % (label, b.last_kernels()[:28], ms, len(b) / ms * 1e3,
                                                             sum(k["waves"] for k in st)), flush=True)
    return ms


if "ab" in what:
    for name, kw in [("uniform 1000 reads, 1 chain, 7500 it", dict(n_reads=1000)),
                     ("uniform 1000 reads, MISO defaults", dict(n_reads=1000, chains=6, iters=5000, burn=500, lag=10)),
                     ("hg19-like reads, 1 chain, 7500 it", dict(n_reads=workload.HG19_LIKE)),
                     ("hg19-like reads, MISO defaults", dict(n_reads=workload.HG19_LIKE, chains=6, iters=5000, burn=500, lag=10))]:
        t0 = time.time()
        b = workload.build_batch(0, E, device_match=True, paired=PAIRED, **kw)
        b.upload(0)
        print("# %s%s (build+upload %.1f s)" % (name, " paired-end" if PAIRED else "", time.time() - t0), flush=True)
        run(b, "  multi (planner)")
        if not PAIRED:
            for w in (8, 4, 1):
                run(b, "  multi, %d wavefronts per workgroup" % w, MISO_K2_WPB=w)
        run(b, "  old (single / two widths)", MISO_K2_MULTI=0)
        if "hg19" in name and "1 chain" in name:
            for G in (8, 64):
                run(b, "  old, %d lanes per chain" % G, MISO_LANES_PER_CHAIN=G)
        del b

if "calib" in what:
    # per wavefront and Gibbs step: ms x SIMD cycles / (wavefront-steps x 4 cycles per VALU) at full occupancy
    iters = 1500
    for reads in (300, 1500):
        b = workload.build_batch(0, 20000, n_reads=reads, chains=6, iters=iters, burn=500, lag=10, device_match=True,
                                 paired=PAIRED)
        b.upload(0)
        for G in ((8, 16, 32, 64) if PAIRED else (1, 2, 3, 4, 8, 16)):
            ms = run(b, "calib reads=%d G=%d" % (reads, G), MISO_LANES_PER_CHAIN=G, MISO_K2_PAIR=0)
            st = b.launch_stats()["kernels"][0]
            valu = ms * 1e-3 * 1024 * 2.4e9 / 4.03 / (st["waves"] * (iters + 1))
            print("    -> %.0f VALU-equivalents per wavefront step, %.1f blocks per lane and step" %
                  (valu, st["trips"] / st["waves"]), flush=True)
        del b
