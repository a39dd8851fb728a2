import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_fuzz as t
from _libs import OrcLib
orc = OrcLib()
bad = 0
for seed in range(5, 45):
    for paired in (False, True):
        try:
            t.test_random_mixed_batches_bit_exact(orc, paired, seed)
        except AssertionError as e:
            bad += 1; print("FAIL seed", seed, "paired", paired, e)
print("done, failures:", bad)
