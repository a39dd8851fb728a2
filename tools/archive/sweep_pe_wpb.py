"""GPU experiment: paired-end two-isoform launches with 4- and 8-wavefront workgroups (MISO_K2W_WPB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

for name, kw in (("uniform 1000 pairs", dict(n_reads=1000)), ("hg19-like", dict(n_reads=workload.HG19_LIKE)),
                 ("hg19-like, MISO defaults", dict(n_reads=workload.HG19_LIKE, chains=6, iters=5000, burn=500, lag=10))):
    b = workload.build_batch(0, 40000, paired=True, device_match=True, **kw)
    b.upload(0)
    print("#", name, flush=True)
    run(b, "  planner's choice")
    run(b, "  4 wavefronts per workgroup", MISO_K2W_WPB=4)
    run(b, "  8 wavefronts per workgroup", MISO_K2W_WPB=8)
    del b
