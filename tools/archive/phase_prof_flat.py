"""GPU experiment: per-phase cycles inside sampler_flat (profile build: tools/build_prof.sh, run with
MISO_AMD_LIB=tools/_build/libmiso_prof.so).  EVENTS, KS (comma list), NCS (comma list of chains per wavefront,
0 = the library's choice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import capi, workload
E, iters = int(os.environ.get("EVENTS", 40000)), 1500
Ks = [int(k) for k in os.environ.get("KS", "5").split(",")]
NCs = [int(g) for g in os.environ.get("NCS", "0").split(",")]
for K in Ks:
    b = workload.build_batch(0, E, K=K, iters=iters, burn=500, device_match=True)
    b.upload(0)
    for nc in NCs:
        if nc:
            os.environ["MISO_FLAT_NC"] = str(nc)
        else:
            os.environ.pop("MISO_FLAT_NC", None)
        b.launch(seed=42); ms = b.sync(); b.download()
        st = b.launch_stats()["kernels"][0]
        acc = np.zeros(3); idx = list(range(0, E, 257))
        for i in idx: acc += b.result(i).loglik[:3]
        acc /= len(idx) * iters
        print("K=%d NC=%s %s %7.1f ms waves %d trips/wave %.1f | cycles/wave-iteration: MH %7.0f thresholds %7.0f read-loop+resolve %7.0f"
              % ((K, nc or "auto", b.last_kernels(), ms, st["waves"], st["trips"] / st["waves"]) + tuple(acc)), flush=True)
