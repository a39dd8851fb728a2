"""GPU: more configurations of tests/test_gpu_fuzz.py -- 40 seeds of the random mixed batches, single- and paired-end
(paired-end also with the buckets of every class in one launch, MISO_PE_MULTI=1, and with every class on sixteen lanes in ONE
launch, sampler_grp_all), and 20 seeds of the collapsed step at both levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as t
from _libs import OrcLib
orc = OrcLib()
bad = 0
for seed in range(5, 45):
    for paired, multi in ((False, False), (True, False), (True, True), (True, "all")):
        for k in ("MISO_PE_MULTI", "MISO_GENERAL_LANES", "MISO_NO_PE_BUCKETS", "MISO_PE_ALL"):
            os.environ.pop(k, None)
        if multi == "all":
            os.environ.update(MISO_GENERAL_LANES="16", MISO_NO_PE_BUCKETS="1", MISO_PE_ALL="1")
        elif multi:
            os.environ["MISO_PE_MULTI"] = "1"
        try:
            t.test_random_mixed_batches_bit_exact(orc, paired, seed)
        except AssertionError as e:
            bad += 1; print("FAIL seed", seed, "paired", paired, "multi", multi, e)
for k in ("MISO_PE_MULTI", "MISO_GENERAL_LANES", "MISO_NO_PE_BUCKETS", "MISO_PE_ALL"):
    os.environ.pop(k, None)
for seed in range(20, 40):
    for level in (1, 2):
        try:
            t.test_random_mixed_batches_collapsed_bit_exact(orc, level, seed)
        except AssertionError as e:
            bad += 1; print("FAIL collapsed level", level, "seed", seed, e)
print("done, failures:", bad)
