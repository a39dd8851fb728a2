import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run
for K, E, reads in ((5, 40000, workload.HG19_LIKE), ((3, 20), 16384, workload.HG19_LIKE), (5, 40000, 1000), ((3, 20), 16384, 1000)):
    b = workload.build_batch(0, E, K=K, paired=True, n_reads=reads, device_match=True, iters=1500, burn=500)
    b.upload(0)
    print("# paired-end K=%s, %d events, %s, 1500 iterations" % (K, E, "hg19-like" if reads != 1000 else "1000 pairs"), flush=True)
    run(b, "  size buckets")
    run(b, "  one launch per class", MISO_NO_PE_BUCKETS=1)
    del b
