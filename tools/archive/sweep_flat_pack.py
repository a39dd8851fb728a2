"""GPU experiment: sampler_flat with wavefronts packed by work units on the hg19-like read counts (K = 5, 3, 10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

for K in (5, 3, 10):
    for name, reads in (("hg19-like", workload.HG19_LIKE), ("uniform 1000", 1000)):
        b = workload.build_batch(0, 40000, K=K, n_reads=reads, device_match=True)
        b.upload(0)
        print("# K=%d %s" % (K, name), flush=True)
        run(b, "  default")
        run(b, "  uniform layout", MISO_FLAT_PACK=0)
        if name.startswith("hg19"):
            for nc in (4, 8, 12, 16):
                run(b, "  packed, nc %d" % nc, MISO_FLAT_NC=nc)
        del b
