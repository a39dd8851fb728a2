"""GPU diagnostic (variant build -DMISO_K2_WAVETIME, tools/build_variant.sh): how long the wavefront of every chain ran,
by the planner's runs of equal lanes per chain -- a flat profile is the goal of plan.cpp's cost model.
    tools/build_variant.sh wavetime "-DMISO_K2_WAVETIME" kernels_k2m_m0w8 kernels_k2m_m0w4 kernels_k2m_m2w4
    MISO_AMD_LIB=tools/_build/libmiso_wavetime.so python tools/wave_time.py [hg19] [paired]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from miso_amd import capi, workload

hg19, paired = "hg19" in sys.argv, "paired" in sys.argv
E = 40000
b = workload.build_batch(0, E, n_reads=workload.HG19_LIKE if hg19 else 1000, paired=paired, device_match=True)
b.upload(0)
b.launch(seed=42); ms = b.sync()
b.launch(seed=42); ms = b.sync()
b.download()
print("# %s, kernel %s, %.2f ms" % ("hg19-like" if hg19 else "uniform", b.last_kernels(), ms))
dur = np.array([int(b.placement(i)[0]) for i in range(E)], dtype=np.float64)
nd = np.zeros(E, np.int64)
for i in range(E):
    t, c = b.classes(i)
    nd[i] = int(sum(cc for tt, cc in zip(t, c) if tt.sum() >= 2))
order = np.argsort(-nd, kind="stable")
# the plan the library made (same inputs through the C ABI)
L = capi.lib()
nds = np.ascontiguousarray(nd[order].astype(np.int32))
n = C.c_int(0); fe = (C.c_int * 17)(); fw = (C.c_int * 17)(); ln = (C.c_int * 16)(); est = (C.c_double * 3)()
L.miso_plan_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
capi.check(L.miso_plan_lanes(nds.ctypes.data, E, 1, int(paired), 512 if paired else 256, 8 if paired else 64, None, 0.0,
                             C.byref(n), fe, fw, ln, est))
d = dur[order] / dur.max()
print("run lanes events wgs | draws first..last | wavefront time / longest: first, mean, last")
for r in range(n.value):
    lo, hi = fe[r], fe[r + 1]
    print("%3d %5d %6d %4d | %6d .. %6d | %.3f %.3f %.3f" % (r, ln[r], hi - lo, fw[r + 1] - fw[r], nds[lo], nds[hi - 1],
                                                          d[lo], d[lo:hi].mean(), d[hi - 1]))
