#!/usr/bin/env python3
"""Kernel time of an arbitrary single-class shape (not only the bench matrix rows):
    python tools/archive/r6_shape.py K=6 K=7,reads=hg19 K=8,paired=1 K=3-20,paired=1,reads=250 [--reps 2] [--events 40000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="+")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--events", type=int, default=40000)
    a = ap.parse_args()
    from miso_amd import capi
    capi.set_device(0)
    for spec in a.shapes:
        ov = {}
        for kv in spec.split(","):
            k, v = kv.split("=")
            if k == "K" and "-" in v:
                ov[k] = tuple(int(x) for x in v.split("-"))   # K=3-20: a whole-gene mix
            elif k == "reads":
                ov[k] = v if v == "hg19" else int(v)
            else:
                ov[k] = bool(int(v)) if k == "paired" else int(v)
        b = bench.build(0, a.events, dict(bench.BASE_SHAPE, **ov))
        b.upload(0)
        ms = []
        for r in range(a.reps + 1):
            b.launch(seed=42, first_event_id=0)
            ms.append(b.sync())
        print("%-24s %-40s median %.2f ms (min %.2f)" % (spec, b.last_kernels()[:40], sorted(ms[1:])[len(ms[1:]) // 2], min(ms[1:])), flush=True)
        del b


if __name__ == "__main__":
    main()
