"""GPU experiment: per-phase cycles inside sampler_grp (profile build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import capi, workload
E, iters = 8192, 1500
for K in (3,):
    b = workload.build_batch(0, E, K=K, iters=iters, burn=500)
    b.upload(0)
    for G in (16,):
        os.environ["MISO_GENERAL_LANES"] = str(G)
        b.launch(seed=42); ms = b.sync(); b.download()
        acc = np.zeros(6); idx = list(range(0, E, 257))
        for i in idx: acc += b.result(i).loglik[:6]
        acc /= len(idx) * iters
        print("K=%d G=%d %7.1f ms | cycles/iter: MH %7.0f thresholds %7.0f read-loop %7.0f philox %7.0f select %7.0f count %7.0f" % ((K, G, ms) + tuple(acc)), flush=True)
