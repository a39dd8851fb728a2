"""Round 6: sampler_flat's packing on hg19-like read counts -- wavefronts (MISO_FLAT_NC = average chains per wavefront, forced) against kernel time, K = 3 ... 12.
    PYTHONPATH=. python tools/archive/flat_pack_sweep6.py [K ...]"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "one":
    K, nc = int(sys.argv[2]), sys.argv[3]
    if nc.startswith("f"):
        os.environ["MISO_FLAT_ROUNDS_FRAC"] = nc[1:]
    elif nc != "-":
        os.environ["MISO_FLAT_NC"] = nc
    os.environ["MISO_TIMING"] = "1"
    from miso_amd import workload
    b = workload.build_batch(0, 40000, K=K, n_reads=workload.HG19_LIKE, device_match=True, iters=1500, burn=500)
    b.upload(0)
    ms = []
    for _ in range(3):
        b.launch(seed=42); ms.append(b.sync())
    print("RESULT K=%d nc=%s %.2f ms %s" % (K, nc, sorted(ms)[1], b.last_kernels()), flush=True)
    sys.exit(0)
for K in [int(x) for x in sys.argv[1:]] or [3, 5, 8, 10]:
    for nc in (os.environ.get("SWEEP_NC") or "-,6,8,10,11,12,13,14,16").split(","):
        out = subprocess.run([sys.executable, __file__, "one", str(K), nc], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        waves = [l for l in out.split("\n") if "flat_waves" in l][-1:]
        res = [l for l in out.split("\n") if l.startswith("RESULT")]
        print((res[0] if res else "no result") + " | " + (waves[0].split("resident")[1].strip() if waves else ""), flush=True)
