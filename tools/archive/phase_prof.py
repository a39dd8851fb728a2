"""GPU experiment: per-phase shader-cycle counts inside sampler_k2 (needs the -DMISO_K2_PROFILE
build: MISO_AMD_LIB=gpurun_out/libmiso_amd_prof.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import capi, workload
E, iters = 40000, 1500
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
b = workload.build_batch(0, E, n_reads=reads, iters=iters, burn=500)
b.upload(0)
for G in (1, 2, 3, 4, 8):
    os.environ["MISO_LANES_PER_CHAIN"] = str(G)
    b.launch(seed=42); ms = b.sync(); b.download()
    acc = np.zeros(5)
    idx = list(range(0, E, 997))
    for i in idx:
        acc += b.result(i).loglik[:5]
    acc /= len(idx) * iters
    print("G=%d  %7.1f ms | cycles/iteration: MH %7.0f  threshold %6.0f  gibbs-loop %7.0f  reduce %6.0f  record %5.0f  | sum %7.0f"
          % ((G, ms) + tuple(acc) + (acc.sum(),)), flush=True)
