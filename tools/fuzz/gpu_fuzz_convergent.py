"""GPU: stop = CONVERGENT_MEAN on random mixed batches -- isoform counts 2 ... 20, 10 ... 900 reads, single- and paired-end and the
collapsed step, schedules short enough that most events need two to four rounds -- every event against the CPU checker's
counter mode (which continues its chains as the reference does; the device re-runs them from the start with the rounds'
openings marked, runtime.hip converge_rounds): samples, log scores, final assignments, accept / reject counts, bit for bit.
    python tools/fuzz/gpu_fuzz_convergent.py [seeds=12]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import miso_amd
from _libs import OrcLib
from _problems import expr_for, flat
from test_gpu_fuzz import random_gene

orc = OrcLib()
bad = total = multi = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for mode in ("se", "pe", "collapsed1", "collapsed2"):
        rng = np.random.default_rng(7000 + seed)
        paired = mode == "pe"
        iters, burn, lag, chains = int(rng.integers(40, 160)), int(rng.integers(0, 30)), int(rng.integers(1, 4)), int(rng.integers(2, 5))
        kw = dict(iters=iters, burn=burn, lag=lag, chains=chains, stop=1, max_iters=int(iters * rng.integers(2, 30)))
        extra = dict(collapsed=int(mode[-1])) if mode.startswith("collapsed") else {}
        b = miso_amd.Batch(36, paired=paired, mean=250.0 if paired else 0.0, var=900.0 if paired else 0.0, **kw, **extra)
        cases = []
        for j in range(14):
            K = int(rng.integers(2, 21))
            exons, isoforms = random_gene(rng, K, 400 if paired else 90, 300 if paired else 80)
            g = orc.gene(flat(exons), isoforms)
            n = int(rng.integers(10, 900))
            orc.rng_seed(100 * seed + j)
            if paired:
                rc, _, pos, cig = orc.simulate_paired_reads(g, expr_for(K), n, 36, 250.0, 900.0)
            else:
                rc, _, pos, cig = orc.simulate_reads(g, expr_for(K), n, 36)
            assert rc == 0
            b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
            cases.append((g, pos, cig, K))
        b.run(seed=seed + 1, first_event_id=500)
        multi += b.rounds() > 1
        omode = OrcLib.COLLAPSED if mode.startswith("collapsed") else OrcLib.COUNTER
        for e, (g, pos, cig, K) in enumerate(cases):
            if mode == "collapsed1" and K > 2:
                omode_e = OrcLib.COUNTER      # level 1: only two-isoform events draw collapsed
            else:
                omode_e = omode
            if paired:
                cpu = orc.miso_paired(g, pos, cig, 36, 250.0, 900.0, mode=omode_e, seed=seed + 1, event_id=500 + e, **kw)
            else:
                cpu = orc.miso(g, pos, cig, 36, mode=omode_e, seed=seed + 1, event_id=500 + e, **kw)
            gpu = b.result(e)
            total += 1
            ok = cpu.rc == 0 and np.array_equal(gpu.samples, cpu.samples) and np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True) \
                and np.array_equal(gpu.assignment, cpu.assignment) and (gpu.rundata.noAccepted, gpu.rundata.noRejected) == (cpu.accepted, cpu.rejected)
            if not ok:
                bad += 1
                print("FAIL seed", seed, mode, "event", e, "K", K, kw, "rounds", b.rounds(), flush=True)
print("done: %d events, %d batches with more than one round, failures: %d" % (total, multi, bad))
sys.exit(1 if bad else 0)
