"""GPU experiment: sampler_k2 throughput versus lanes-per-chain G."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import capi, workload

E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 7500
reads = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
b = workload.build_batch(0, E, n_reads=reads, iters=iters, burn=iters // 3)
b.upload(0)
os.environ.pop("MISO_LANES_PER_CHAIN", None)
b.launch(seed=42); ms = b.sync(); b.launch(seed=42); ms = b.sync()
print("auto      %8.1f ms  %9.0f events/s" % (ms, E / ms * 1e3), flush=True)
for G in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32, 64):
    os.environ["MISO_LANES_PER_CHAIN"] = str(G)
    b.launch(seed=42); ms = b.sync()
    b.launch(seed=42); ms = b.sync()
    print("G=%2d  %8.1f ms  %9.0f events/s" % (G, ms, E / ms * 1e3), flush=True)
