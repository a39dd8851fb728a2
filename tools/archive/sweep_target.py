"""GPU experiment: kernel time of the headline batch against the planner's bound on a wavefront's step (MISO_K2_TARGET)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

reads = workload.HG19_LIKE if (len(sys.argv) > 1 and sys.argv[1] == "hg19") else 1000
b = workload.build_batch(0, 40000, n_reads=reads, device_match=True)
b.upload(0)
run(b, "planner")
for D in (2600, 2800, 3000, 3200, 3400, 3600, 3800, 4000, 4200, 4400, 4700, 5000, 5500, 6000):
    run(b, "target %d" % D, MISO_K2_TARGET=D)
run(b, "old", MISO_K2_MULTI=0)
