#!/usr/bin/env python3
"""Where did sampler_k2's wavefronts run?  (GPU; tools/placement_check.py [events])
Runs a short two-isoform batch, reads every chain's HW_REG_HW_ID back and reports, for the paired
launch (8 wavefronts per workgroup), whether the heavy wavefront p and the light wavefront W-1-p of
each pair really shared a SIMD, and how the workgroups spread over XCDs / CUs."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import capi, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
capi.set_device(0)
b = workload.build_batch(0, n, K=2, n_reads=1000, iters=40, burn=10, lag=1, chains=1, device_match=True)
b.upload(0)
b.launch(seed=1, first_event_id=0)
b.sync()
b.download()
name = b.last_kernels()
G = int(name.split("<")[1].split(",")[0])
cpw = 64 // G
nd = []
for i in range(n):
    t, c = b.classes(i)
    nd.append(int(sum(cc for tt, cc in zip(t, c) if tt.sum() >= 2)))
order = sorted(range(n), key=lambda i: -nd[i])          # stable, like the library's sort
hw = np.array([int(b.placement(i)[0]) for i in range(n)], dtype=np.uint64)
waves = [order[j:j + cpw] for j in range(0, n, cpw)]
W = len(waves)
def fields(h):
    h = int(h)
    return {"wave": h & 15, "simd": (h >> 4) & 3, "pipe": (h >> 6) & 3, "cu": (h >> 8) & 15, "sh": (h >> 12) & 1,
            "se": (h >> 13) & 7, "tg": (h >> 16) & 15}
wave_hw = []
for w in waves:
    ids = set(int(hw[i]) for i in w)
    wave_hw.append(ids.pop() if len(ids) == 1 else None)
print(name, "waves", W, "mixed-id waves", sum(1 for x in wave_hw if x is None))
Wp = (W + 7) // 8 * 8
same = diff = 0
for p in range(Wp // 2):
    a, c = p, Wp - 1 - p
    if c >= W or wave_hw[a] is None or wave_hw[c] is None:
        continue
    fa, fc = fields(wave_hw[a]), fields(wave_hw[c])
    key = lambda f: (f["se"], f["sh"], f["cu"], f["simd"])
    if key(fa) == key(fc):
        same += 1
    else:
        diff += 1
print("pairs on the same SIMD:", same, "on different SIMDs:", diff)
per_simd = collections.Counter()
for x in wave_hw:
    if x is not None:
        f = fields(x)
        per_simd[(f["se"], f["sh"], f["cu"], f["simd"])] += 1
print("distinct (se, sh, cu, simd):", len(per_simd), "waves per SIMD histogram:", collections.Counter(per_simd.values()))
for j in list(range(0, 10)) + list(range(W - 4, W)):
    print(j, "n_draw", nd[waves[j][0]], fields(wave_hw[j]) if wave_hw[j] is not None else None)
