"""Child process of tests/test_gpu_heavy_tail.py::test_two_processes_on_one_gpu...: samples a heavy-tailed batch whose
largest events run as chains on SEVERAL workgroups (coop.hpp) and checks every output against the oracle's counter mode.
Prints one line: "ok retries=<n> kernels=<...>".  argv: paired(0/1) K rounds"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import miso_amd                                        # noqa: E402
from _libs import OrcLib                               # noqa: E402
from _problems import flat, se_gene, expr_for          # noqa: E402


def main():
    paired, K, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    orc = OrcLib()
    sizes = [30, 22000, 200, 5, 0, 1800, 9000, 60, 700, 120, 90, 200, 35, 400, 150, 80] if K > 2 else \
        [20, 30000, 300, 5, 0, 2500, 20000, 40, 1000, 150, 7000, 64, 16000, 3, 511]
    kw = dict(iters=60, burn=10, lag=2, chains=1)
    evs, cpu = [], []
    for j, n in enumerate(sizes):
        exons, isoforms = se_gene(K, exlen=(500 if paired else 90) + 7 * j, gap=300 if paired else 100)
        g = orc.gene(flat(exons), isoforms)
        orc.rng_seed(7000 + j)
        if paired:
            rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(K), max(n, 1), 36, 250.0, 900.0)
            pos, cig = pos[:2 * n], cig[:2 * n]
            r = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=17, event_id=300 + j, trace=True, **kw)
        else:
            rc, _, pos, cig = orc.simulate_reads(g, expr_for(K), max(n, 1), 36)
            pos, cig = pos[:n], cig[:n]
            r = orc.miso(g, pos, cig, 36, mode=OrcLib.COUNTER, seed=17, event_id=300 + j, trace=True, **kw)
        assert rc == 0 and r.rc == 0
        evs.append((exons, isoforms, pos, cig))
        cpu.append(r)
    retries = 0
    kernels = ""
    for _ in range(rounds):
        b = miso_amd.Batch(36, paired=bool(paired), mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0, device_match=True, **kw)
        for exons, isoforms, pos, cig in evs:
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        b.run(seed=17, first_event_id=300)
        retries += b.coop_retries()
        kernels = b.last_kernels()
        for i, r in enumerate(cpu):
            gpu = b.result(i)
            assert (gpu.counts_hash == r.trace["counts_hash"]).all(), (i, sizes[i])
            assert np.array_equal(gpu.samples, r.samples, equal_nan=True), (i, sizes[i])
            assert np.array_equal(gpu.loglik, r.loglik, equal_nan=True), (i, sizes[i])
            assert (gpu.assignment == r.assignment).all(), (i, sizes[i])
            assert gpu.rundata.noAccepted == r.accepted, (i, sizes[i])
    print("ok retries=%d kernels=%s" % (retries, kernels), flush=True)


if __name__ == "__main__":
    main()
