"""GPU experiment: sampler_k2 throughput versus lanes-per-chain G and register budget W."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import capi, workload

E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 7500
b = workload.build_batch(0, E, iters=iters, burn=iters // 3)
b.upload(0)
for W in (2, 3, 4):
    for G in (1, 2, 4, 8, 16, 64):
        os.environ["MISO_K2_WAVES"] = str(W)
        os.environ["MISO_LANES_PER_CHAIN"] = str(G)
        b.launch(seed=42); ms = b.sync()
        b.launch(seed=42); ms = b.sync()
        print("W=%d G=%2d  %8.1f ms  %9.0f events/s" % (W, G, ms, E / ms * 1e3), flush=True)
