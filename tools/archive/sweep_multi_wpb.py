"""GPU experiment: wavefronts per workgroup of sampler_k2_multi when the launch needs several rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

for name, kw in [("uniform 1000 reads, MISO defaults", dict(n_reads=1000, chains=6, iters=5000, burn=500, lag=10)),
                 ("hg19-like reads, MISO defaults", dict(n_reads=workload.HG19_LIKE, chains=6, iters=5000, burn=500, lag=10))]:
    b = workload.build_batch(0, 40000, device_match=True, **kw)
    b.upload(0)
    print("#", name, flush=True)
    run(b, "  multi (planner)")
    for w in (8, 4, 1):
        run(b, "  multi, %d wavefronts per workgroup" % w, MISO_K2_WPB=w)
        run(b, "  multi, %d per workgroup, narrowest layout" % w, MISO_K2_WPB=w, MISO_K2_TARGET="1e12")
    run(b, "  old (single / two widths)", MISO_K2_MULTI=0)
