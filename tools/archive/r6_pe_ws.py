#!/usr/bin/env python3
"""Round 6: is the paired-end K >= 3 read loop waiting for its records?  The same number of chains (wavefronts, instructions)
over fewer EVENTS: chains of one event read the same records, so the launch's working set shrinks from 240 MB (beyond the
L2s, every record load a trip to the infinity cache / HBM) to 24 and 2.4 MB (L2-resident) while nothing else changes.
    python tools/r6_pe_ws.py [K ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402


def main():
    from miso_amd import capi
    capi.set_device(0)
    for K in [int(x) for x in sys.argv[1:]] or [5, 10]:
        total = 40000 if K < 10 else 20000
        # the same kernel and lanes per chain whatever the events' share of the batch (the size buckets would give a gene
        # that is a hundredth of the batch a wavefront of its own)
        os.environ["MISO_NO_PE_BUCKETS"] = "1"
        os.environ["MISO_GENERAL_LANES"] = "8" if K < 10 else "16"
        for events, chains in ((total, 1), (total // 10, 10), (total // 100, 100)):
            sh = dict(bench.BASE_SHAPE, K=K, paired=True, chains=chains)
            b = bench.build(0, events, sh)
            b.upload(0)
            ms = []
            for r in range(3):
                b.launch(seed=42, first_event_id=0)
                ms.append(b.sync())
            print("K=%d paired: %6d events x %3d chains  kernel %8.2f ms  (%s)" % (K, events, chains, sorted(ms[1:])[0], b.last_kernels()), flush=True)
            del b


if __name__ == "__main__":
    main()
