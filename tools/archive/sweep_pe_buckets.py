"""GPU experiment (round 3): where a paired-end gene moves from 32 lanes to a wavefront of its own (MISO_PE_T_WAVE)
and from there to a workgroup (MISO_PE_T_WIDE), in units of the work-share rule of runtime.hip upload() -- the
lanes a gene needs to finish in half the batch's ideal time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

cases = [((3, 20), 16384), (5, 40000), (10, 20000)]
for K, E in cases:
    for tw, td in ((48, 48), (48, 96), (48, 128), (48, 192), (48, 256), (32, 128), (64, 192), (96, 256)):
        os.environ["MISO_PE_T_WAVE"] = str(tw); os.environ["MISO_PE_T_WIDE"] = str(td)
        b = workload.build_batch(0, E, K=K, paired=True, n_reads=workload.HG19_LIKE, device_match=True, iters=1500, burn=500)
        b.upload(0)
        run(b, "K=%s hg19-like: a wavefront from %d, a workgroup from %d" % (K, tw, td))
        del b
