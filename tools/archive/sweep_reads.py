"""GPU experiment: separate the per-iteration scalar (MH) cost from the per-read (Gibbs) cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import capi, workload
E, iters = 40000, 1500
for G in (4, 8):
    os.environ["MISO_LANES_PER_CHAIN"] = str(G)
    for reads in (8, 250, 1000, 4000):
        b = workload.build_batch(0, E, n_reads=reads, iters=iters, burn=500)
        b.upload(0)
        b.launch(seed=1); b.sync(); b.launch(seed=1); ms = b.sync()
        r = b.result(0) if False else None
        print("G=%d reads=%5d  %8.2f ms  -> %7.1f ns per chain-iteration" % (G, reads, ms, ms * 1e6 / (E * iters)), flush=True)
        del b
