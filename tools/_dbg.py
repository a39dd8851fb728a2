import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run
for rep in range(2):
  for tw, td in ((48, 48), (48, 256), (64, 192), (96, 256), (64, 256), (64, 384)):
    os.environ["MISO_PE_T_WAVE"] = str(tw); os.environ["MISO_PE_T_WIDE"] = str(td)
    b = workload.build_batch(0, 16384, K=(3, 20), paired=True, n_reads=workload.HG19_LIKE, device_match=True, iters=1500, burn=500)
    b.upload(0)
    run(b, "q=%s: a wavefront from %d, a workgroup from %d" % (os.environ.get("GPU_MAX_HW_QUEUES"), tw, td))
    del b
