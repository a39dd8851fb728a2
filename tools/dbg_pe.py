import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import miso_amd
from _libs import OrcLib
from _problems import flat
from miso_amd import workload
orc = OrcLib()
for K in (3, 5, 10):
    exons, isoforms, pos, cig = workload.event_reads(7, K, 600, paired=True)
    g = orc.gene(flat(exons), isoforms)
    kw = dict(iters=300, burn=100, lag=2, chains=3)
    b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, **kw)
    for _ in range(5): b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
    b.run(seed=5, first_event_id=0)
    for e in range(5):
        cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=e, **kw)
        r = b.result(e)
        bad = np.nonzero(r.loglik != cpu.loglik)[0]
        print("K", K, "event", e, b.last_kernels(), "samples equal", np.array_equal(r.samples, cpu.samples), "loglik mismatches", len(bad), "cols mod 3:", sorted(set(int(x) % 3 for x in bad)), (r.loglik[bad[:3]] - cpu.loglik[bad[:3]]) if len(bad) else "")
