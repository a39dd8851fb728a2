import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import miso_amd
from _libs import OrcLib
from _problems import flat
from miso_amd import workload
orc = OrcLib()
for n, nev in ((40, 1), (40, 7), (3, 2), (400, 3), (1000, 64)):
    exons, isoforms, pos, cig = workload.event_reads(7, 2, n, paired=True)
    g = orc.gene(flat(exons), isoforms)
    kw = dict(iters=60, burn=20, lag=2, chains=3)
    b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, counts_trace=True, **kw)
    for _ in range(nev): b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=5, first_event_id=0)
    for e in range(min(nev, 3)):
        cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=e, trace=True, **kw)
        r = b.result(e, trace=True)
        ct, cc = r.counts_trace, cpu.trace["counts_trace"]
        bad = np.argwhere(ct != cc)
        print("pairs", n, "events", nev, "event", e, b.last_kernels(), "trace mismatches", len(bad), bad[:2].tolist(), ct.reshape(-1)[:6], cc.reshape(-1)[:6])
