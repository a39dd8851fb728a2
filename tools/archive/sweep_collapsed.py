"""GPU experiment (round 3): the collapsed Gibbs step (miso_batch_set_collapsed) against the per-read sweep, and its
lanes per chain (MISO_COLLAPSED_LANES: 1 = sampler_lane, 2 / 4 / 8 = sampler_k2c)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

cfgs = [("1 chain x 7500 iterations", dict()), ("MISO defaults: 6 chains x 5000 iterations", dict(chains=6, iters=5000, burn=500, lag=10))]
for nr, name in ((1000, "1000 reads"), (workload.HG19_LIKE, "hg19-like read counts")):
    for cname, kw in cfgs:
        print("# single-end K=2, 40 000 events, %s, %s" % (name, cname), flush=True)
        b = workload.build_batch(0, 40000, n_reads=nr, device_match=True, **kw)
        b.upload(0)
        run(b, "  per-read sweep (default)")
        del b
        b = workload.build_batch(0, 40000, n_reads=nr, device_match=True, collapsed=True, **kw)
        b.upload(0)
        run(b, "  collapsed, lanes per chain by rule")
        for g in (1, 2, 4, 8):
            run(b, "  collapsed, %d lane(s) per chain" % g, MISO_COLLAPSED_LANES=g)
        del b
if "k" in sys.argv:
    for K in (3, 5, 10):
        for nr, name in ((1000, "1000 reads"), (10000, "10 000 reads"), (workload.HG19_LIKE, "hg19-like")):
            E = 40000 if nr != 10000 else 8000
            print("# single-end K=%d, %d events, %s, 1500 iterations" % (K, E, name), flush=True)
            for coll in (0, 2):
                b = workload.build_batch(0, E, K=K, n_reads=nr, device_match=True, collapsed=coll, iters=1500, burn=500)
                b.upload(0)
                run(b, "  collapsed (sampler_lane_k)" if coll else "  per-read sweep (default)")
                del b
