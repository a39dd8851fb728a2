"""CPU: the lanes-per-chain planner of the two-isoform kernels (miso_amd/csrc/plan.cpp through miso_plan_lanes).

The reference's cost per event is O(reads) and events share nothing (miso.c:845-900); the planner must turn ANY
list of read counts into runs of equal lanes per chain in which no wavefront outlasts the launch.  Host arithmetic
only -- no GPU call.
"""
import ctypes as C

import numpy as np
import pytest

from miso_amd import capi, workload

WIDE = 512


def plan(nd, chains=1, paired=0, resident=256, max_cpw=64, cost=None, target=0.0):
    L = capi.lib()
    nd = np.ascontiguousarray(np.asarray(nd, dtype=np.int32))
    n = C.c_int(0)
    fe, fw, ln, est = (C.c_int * 17)(), (C.c_int * 17)(), (C.c_int * 16)(), (C.c_double * 3)()
    c5 = None if cost is None else (C.c_double * 5)(*cost)
    L.miso_plan_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    capi.check(L.miso_plan_lanes(nd.ctypes.data, len(nd), chains, paired, resident, max_cpw, c5, target,
                                 C.byref(n), fe, fw, ln, est))
    runs = [dict(first=fe[i], events=fe[i + 1] - fe[i], wgs=fw[i + 1] - fw[i], lanes=ln[i]) for i in range(n.value)]
    return runs, dict(total=est[0], longest=est[1], rounds=int(est[2]), wgs=fw[n.value], events=fe[n.value])


def draws(n_reads_spec, n=40000, frac=0.65):
    r = np.array([workload.event_n_reads(e, n_reads_spec) for e in range(n)])
    return np.sort((r * frac).astype(np.int32))[::-1]


def check_shape(runs, info, nd, chains, wpb):
    assert info["events"] == len(nd)
    assert [r["first"] for r in runs] == list(np.cumsum([0] + [r["events"] for r in runs[:-1]]))
    lanes = [r["lanes"] for r in runs]
    assert lanes == sorted(lanes, reverse=True) and len(set(lanes)) == len(lanes)   # wider for larger events, one run per width
    for r in runs:
        ch = r["events"] * chains
        if r["lanes"] == WIDE:      # one workgroup per chain, or several for the very largest (coop.hpp)
            assert ch <= r["wgs"] <= 32 * ch, r
        else:
            assert r["wgs"] == -(-(-(-ch // (64 // r["lanes"]))) // wpb), r


def test_uniform_batch_fills_the_device_in_one_round():
    nd = np.sort(np.random.default_rng(0).integers(127, 940, 40000).astype(np.int32))[::-1]
    runs, info = plan(nd)
    check_shape(runs, info, nd, 1, 8)
    assert info["rounds"] == 1 and 0.9 * 256 <= info["wgs"] <= 256
    assert all(1 <= r["lanes"] <= 8 for r in runs)


def test_heavy_tail_no_wavefront_outlasts_the_launch():
    nd = draws(workload.HG19_LIKE)
    assert nd[0] > 50000 and nd[-1] < 30
    for chains, wpb in ((1, 8), (6, 8)):
        runs, info = plan(nd, chains=chains)
        check_shape(runs, info, nd, chains, wpb)
        assert runs[0]["lanes"] in (64, WIDE) and runs[-1]["lanes"] <= 2   # (a whole workgroup only when 64 lanes overshoot the bound)
        # time of the launch >= total work / SIMDs that have work; a wavefront shares its SIMD with another one and
        # advances at half speed, so twice the longest wavefront must stay near that share ("kernel duration <= 2 x
        # the mean" would be 2.0 here)
        simds = 1024 if info["rounds"] > 1 else min(1024, info["wgs"] * 4)
        assert 2 * info["longest"] <= (1.15 if info["rounds"] > 1 else 1.35) * info["total"] / simds, info
    # one width for everybody (what the single-width launch does) would be >= 20 x worse
    one, info1 = plan(nd, target=1e12)
    assert len(one) == 1 and info1["longest"] > 20 * info["longest"]


def test_small_and_degenerate_batches():
    runs, info = plan([36000, 500, 300, 50, 20, 0, 0])
    check_shape(runs, info, [0] * 7, 1, 8)
    assert runs[0]["lanes"] == WIDE and runs[-1]["lanes"] <= 2 and info["rounds"] == 1
    runs, info = plan([0])
    assert len(runs) == 1 and runs[0]["wgs"] == 1
    runs, info = plan([], chains=3)
    assert runs == []
    with pytest.raises(capi.InternalError):
        plan([10, 20])          # must be ordered, most drawing reads first


def test_paired_end_respects_the_lds_limit():
    nd = draws(workload.HG19_LIKE, frac=0.45)
    runs, info = plan(nd, paired=1, resident=512, max_cpw=8)
    check_shape(runs, info, nd, 1, 4)
    assert all(r["lanes"] == WIDE or 64 // r["lanes"] <= 8 for r in runs)
    assert runs[0]["lanes"] == WIDE and runs[-1]["lanes"] == 8
    runs, info = plan(nd, chains=6, paired=1, resident=512, max_cpw=16)    # several rounds: as narrow as the LDS allows
    assert info["rounds"] == 2 and runs[-1]["lanes"] == 4


def test_forced_target_gives_many_widths_on_a_small_batch():
    nd = np.sort(np.array([20, 60000, 300, 5, 2500, 20000, 40, 1000, 150, 7000], dtype=np.int32))[::-1]
    wide, _ = plan(nd, target=1400.0)
    narrow, _ = plan(nd, target=1e12)
    assert len(wide) >= 4 and wide[0]["lanes"] == WIDE
    assert len(narrow) == 1 and narrow[0]["lanes"] == 1
