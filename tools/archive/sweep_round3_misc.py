"""GPU experiments (round 3): (a) paired-end K >= 3 with 8 lanes per chain (8 chains per wavefront share the scalar
step) against the rule's 16 / 32; (b) sampler_flat with more chains per wavefront under the walking read loop (lanes
own contiguous unit ranges: balanced for any chain count) against the descriptor loop's 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miso_amd import workload
from tools.sweep_multi import run

what = sys.argv[1:] or ["pe", "flat"]
if "pe" in what:
    for K, E in ((3, 40000), (5, 40000), (10, 20000)):
        b = workload.build_batch(0, E, K=K, paired=True, device_match=True)
        b.upload(0)
        print("# paired-end K=%d, %d events" % (K, E), flush=True)
        run(b, "  rule")
        for G in (4, 8, 16, 32):
            run(b, "  %d lanes per chain" % G, MISO_GENERAL_LANES=G)
        del b
if "flat" in what:
    for K in (5, 7, 10):
        b = workload.build_batch(0, 40000, K=K, device_match=True)
        b.upload(0)
        print("# single-end K=%d" % K, flush=True)
        run(b, "  rule (descriptor loop)")
        for nc in (6, 8, 10, 12, 14):
            run(b, "  descriptor loop, %d chains per wavefront" % nc, MISO_FLAT_NC=nc)
            run(b, "  walking loop,    %d chains per wavefront" % nc, MISO_FLAT_NC=nc, MISO_FLAT_NO_DESC=1)
        del b
