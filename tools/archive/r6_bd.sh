#!/bin/bash
# round 6: the collapsed kernels (kernels_lane) without machine-level hoisting
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bd; mkdir -p $O
cat > $O/rows.py <<'P'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from miso_amd import capi
capi.set_device(0)
for label, ov in (("main_collapsed", {}), ("se_k2_defaults_collapsed", dict(chains=6, iters=5000, burn=500, lag=10)),
                  ("se_k2_hg19_collapsed", dict(reads="hg19")), ("se_k2_hg19_defaults_collapsed", dict(reads="hg19", chains=6, iters=5000, burn=500, lag=10))):
    b = bench.build(0, 40000, dict(bench.BASE_SHAPE, **ov), collapsed=True)
    b.upload(0)
    ms = []
    for r in range(5):
        b.launch(seed=42, first_event_id=0); ms.append(b.sync())
    print("%-32s %-20s median %.3f ms" % (label, b.last_kernels(), sorted(ms[1:])[2]), flush=True)
P
for lib in miso_amd/libmiso_amd.so tools/_build/libmiso_lanenolicm.so miso_amd/libmiso_amd.so tools/_build/libmiso_lanenolicm.so; do
  echo "== $lib" >> $O/ab.txt
  MISO_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python $O/rows.py 2>&1 | grep median >> $O/ab.txt
done
cat $O/ab.txt
