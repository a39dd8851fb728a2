"""GPU experiment: sampler_grp lanes-per-chain sweep at full batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
E = int(os.environ.get("EVENTS", 40000)); iters = int(os.environ.get("ITERS", 1500))
paired = bool(int(os.environ.get("PAIRED", "0")))
for K in [int(k) for k in os.environ.get("KS", "3").split(",")]:
    b = workload.build_batch(0, E, K=K, iters=iters, burn=500, paired=paired)
    b.upload(0)
    for G in [int(g) for g in os.environ.get("GS", "2,4,8,16").split(",")]:
        os.environ["MISO_GENERAL_LANES"] = str(G)
        b.launch(seed=42); ms = b.sync()
        print("K=%d E=%d G=%d %s %8.1f ms  -> %.0f events/s at 7500 iters" % (K, E, G, b.last_kernels(), ms, E / (ms * 1e-3) * iters / 7500), flush=True)
