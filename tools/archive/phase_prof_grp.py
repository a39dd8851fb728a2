"""GPU experiment: per-phase cycles inside sampler_grp (profile build: tools/build_prof.sh, run with
MISO_AMD_LIB=tools/_build/libmiso_prof.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import capi, workload
E, iters = int(os.environ.get("EVENTS", 8192)), 1500
Ks = [int(k) for k in os.environ.get("KS", "3").split(",")]
Gs = [int(g) for g in os.environ.get("GS", "16").split(",")]
for K in Ks:
    b = workload.build_batch(0, E, K=K, iters=iters, burn=500, paired=bool(int(os.environ.get("PAIRED", "0"))))
    b.upload(0)
    for G in Gs:
        os.environ["MISO_GENERAL_LANES"] = str(G)
        b.launch(seed=42); ms = b.sync(); b.download()
        acc = np.zeros(3); idx = list(range(0, E, 257))
        for i in idx: acc += b.result(i).loglik[:3]
        acc /= len(idx) * iters
        print("K=%d G=%d %s %7.1f ms | cycles/iter: MH %7.0f thresholds %7.0f read-loop+resolve %7.0f" % ((K, G, b.last_kernels(), ms) + tuple(acc)), flush=True)
