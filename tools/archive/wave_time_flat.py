"""GPU diagnostic (variant build -DMISO_FLAT_WAVETIME): single-end K >= 3, when the wavefront of every chain started and how long it
ran -- which events bound the launch, how full the device is over the launch.
    tools/build_variant.sh flatwt "-DMISO_FLAT_WAVETIME" kernels_flat_c8
    MISO_AMD_LIB=tools/_build/libmiso_flatwt.so PYTHONPATH=. python tools/archive/wave_time_flat.py [uniform] [K]"""
import os, sys
import numpy as np
from miso_amd import workload

E = 40000
K = int([x for x in sys.argv[1:] if x.isdigit()][0]) if any(x.isdigit() for x in sys.argv[1:]) else 5
nr = 1000 if "uniform" in sys.argv else workload.HG19_LIKE
b = workload.build_batch(0, E, K=K, n_reads=nr, device_match=True, iters=1500, burn=500)
b.upload(0)
b.launch(seed=42); ms = b.sync()
b.launch(seed=42); ms = b.sync()
b.download()
print("# kernel %s, %.2f ms, K = %d, %s" % (b.last_kernels()[:120], ms, K, "1000 reads" if nr == 1000 else "hg19-like read counts"))
dur = np.array([int(b.placement(i)[0]) for i in range(E)], dtype=np.float64) / 1e5   # ms
t0 = np.array([int(b.result(i).counts_hash[0]) for i in range(E)], dtype=np.float64) / 1e5
t0 -= t0.min()
t1 = t0 + dur
end = t1.max()
nd = np.array([workload.event_n_reads(i, nr) for i in range(E)])
print("# chains running at t (ms), and how many of them belong to events of >= 4000 reads")
for t in np.arange(0, end, end / 24):
    on = (t0 <= t) & (t1 > t)
    print("%7.1f %6d %5d" % (t, int(on.sum()), int((on & (nd >= 4000)).sum())))
order = np.argsort(-nd, kind="stable")
edges = [0, 1, 4, 16, 64, 256, 1024, 4096, 16384, E]
print("reads from..to   events | start ms: min max | chain time ms: min mean max | end ms max")
for lo, hi in zip(edges[:-1], edges[1:]):
    s = order[lo:hi]
    print("%7d..%-7d %6d | %7.1f %7.1f | %7.2f %7.2f %7.2f | %7.1f" % (nd[s].max(), nd[s].min(), len(s), t0[s].min(), t0[s].max(),
                                                                       dur[s].min(), dur[s].mean(), dur[s].max(), t1[s].max()))
