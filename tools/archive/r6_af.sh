#!/bin/bash
# round 6: the whole-gene mix is exactly two rounds of sixteen-lane wavefronts (4096 on 2048 slots): eight lanes for the small classes?
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6af; mkdir -p $O
run() { echo "== $*" >> $O/mix_lanes.txt; env "$@" timeout 600 python tools/r6_clock.py pe_mix pe_mix_hg19 --reps 4 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-170 >> $O/mix_lanes.txt; }
run MISO_X=0
run MISO_PE_LANES8=1
run MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:8,12:16,16:16,32:16
run MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:16,12:16,16:16,32:16
run MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:16
cat $O/mix_lanes.txt
