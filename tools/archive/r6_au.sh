#!/bin/bash
# round 6: sampler_grp_all, the order of its segments: classes longest first (0), a piece per isoform count longest first (1), heaviest / lightest alternating (2), lightest first (3)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6au; mkdir -p $O
for v in 0 1 2 3 0 1 2 3; do
  echo "== MISO_PE_ALL_ORDER=$v" >> $O/ab.txt
  MISO_PE_ALL_ORDER=$v timeout 900 python tools/r6_clock.py pe_mix --reps 4 --probe 0 2>&1 | grep -E "median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
