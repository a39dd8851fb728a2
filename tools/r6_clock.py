#!/usr/bin/env python3
"""Round 6: kernel time x shader clock of a bench row, launch after launch on one box (is a slow
`se_k2_defaults` a slow box or a slow layout?).  Every launch carries the library's clock probe.
    python tools/r6_clock.py [row ...] [--reps 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rows", nargs="*", default=["main", "se_k2_defaults"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--probe", type=int, default=1)
    a = ap.parse_args()
    from miso_amd import capi
    capi.set_device(0)
    shapes = {"main": (dict(bench.BASE_SHAPE), 40000)}
    for wid, _, ov, n, _ in bench.MATRIX:
        shapes[wid] = (dict(bench.BASE_SHAPE, **ov), n)
    for wid in a.rows:
        sh, n = shapes[wid]
        b = bench.build(0, n, sh)
        b.upload(0)
        b.set_clock_probe(bool(a.probe))
        out = []
        for r in range(a.reps + 1):
            t0 = time.perf_counter()
            b.launch(seed=42, first_event_id=0)
            ms = b.sync()
            wall = 1e3 * (time.perf_counter() - t0)
            ghz, win = b.last_clock()
            if r:
                out.append((ms, ghz, win, wall))
        print("%s  kernels %s" % (wid, b.last_kernels()))
        for ms, ghz, win, wall in out:
            print("  kernel %8.3f ms  clock %.4f GHz  window %8.3f ms  Mcycles %9.2f  wall %8.2f ms  -> %7.1f k events/s"
                  % (ms, ghz, win, ms * ghz * 1e3, wall, n / wall))
        ks = sorted(o[0] for o in out)
        print("  median kernel %.3f ms, min %.3f, max %.3f; clock %.4f .. %.4f GHz" % (
            ks[len(ks) // 2], ks[0], ks[-1], min(o[1] for o in out), max(o[1] for o in out)), flush=True)
        del b


if __name__ == "__main__":
    main()
