"""GPU diagnostic (variant build -DMISO_GRP_WAVETIME): how long the wavefront of every chain of a paired-end gene
batch ran, per isoform-count class and size decile -- which genes bound the launch.
    tools/build_variant.sh grpwt "-DMISO_GRP_WAVETIME" kernels_grp_c4 kernels_grp_c8 kernels_grp_c12 kernels_grp_c16 kernels_grp_c32
    MISO_AMD_LIB=tools/_build/libmiso_grpwt.so python tools/wave_time_grp.py [uniform]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from miso_amd import workload

E = int(os.environ.get("WT_E", "16384"))
nr = 1000 if "uniform" in sys.argv else workload.HG19_LIKE
KK = int(os.environ["WT_K"]) if os.environ.get("WT_K") else (3, 20)   # WT_K=5 WT_E=40000: one isoform count
b = workload.build_batch(0, E, K=KK, paired=True, n_reads=nr, device_match=True, iters=1500, burn=500)
b.upload(0)
b.launch(seed=42); ms = b.sync()
b.launch(seed=42); ms = b.sync()
b.download()
print("# kernel %s, %.2f ms" % (b.last_kernels()[:200], ms))
dur = np.array([int(b.placement(i)[0]) for i in range(E)], dtype=np.float64) / 1e5   # ms
K = np.zeros(E, np.int64)
nd = np.zeros(E, np.int64)
for i in range(E):
    t, c = b.classes(i)
    K[i] = t.shape[1]
    nd[i] = int(sum(cc for tt, cc in zip(t, c) if tt.sum() >= 2))
kc = np.where(K <= 4, 4, np.where(K <= 8, 8, np.where(K <= 12, 12, np.where(K <= 16, 16, 32))))
if "timeline" in sys.argv:   # how many chains of every class are running over time
    t0 = np.array([int(b.result(i).counts_hash[0]) for i in range(E)], dtype=np.float64) / 1e5
    t0 -= t0.min()
    t1 = t0 + dur
    end = t1.max()
    print("# wavefront-chains running at t (ms): per class, 32 lanes => 2 chains per wavefront")
    print("   t   " + " ".join("kc%-4d" % c for c in (32, 16, 12, 8, 4)) + "  all   pairs of the chains running: median / max")
    for t in np.arange(0, end, end / 24):
        row = [int(((t0 <= t) & (t1 > t) & (kc == c)).sum()) for c in (32, 16, 12, 8, 4)]
        on = (t0 <= t) & (t1 > t)
        print("%6.1f " % t + " ".join("%6d" % r for r in row) + " %6d" % sum(row) + ("   %8d %8d" % (np.median(nd[on]), nd[on].max()) if on.any() else ""))
print("class  draws from..to   genes | chain time ms: min mean max")
for c in (32, 16, 12, 8, 4):
    idx = np.where(kc == c)[0]
    idx = idx[np.argsort(-nd[idx], kind="stable")]
    edges = [0, 1, 4, 16, 64, 256, 1024, len(idx)]
    for lo, hi in zip(edges[:-1], edges[1:]):
        if lo >= len(idx): break
        s = idx[lo:hi]
        print("%5d %7d..%-7d %6d | %8.2f %8.2f %8.2f" % (c, nd[s].max(), nd[s].min(), len(s), dur[s].min(), dur[s].mean(), dur[s].max()))
