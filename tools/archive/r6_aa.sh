#!/bin/bash
# round 6: how sampler_flat packs the hg19-like K = 5 batch, and other packings (MISO_FLAT_NC / MISO_FLAT_WGS / MISO_PRIO_QUARTILES)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aa; mkdir -p $O
run() { echo "== $*" >> $O/flat_pack.txt; env MISO_TIMING=1 "$@" timeout 300 python tools/r6_clock.py se_k5_hg19 --reps 3 --probe 0 2>&1 | grep -E "flat_waves|median" | sort -u >> $O/flat_pack.txt; }
run MISO_X=0
run MISO_FLAT_NC=10
run MISO_FLAT_NC=12
run MISO_FLAT_NC=13
run MISO_FLAT_NC=14
run MISO_FLAT_WGS=2
run MISO_FLAT_WGS=2 MISO_FLAT_NC=16
run MISO_FLAT_WGS=2 MISO_FLAT_NC=20
run MISO_PRIO_QUARTILES=1
cat $O/flat_pack.txt
