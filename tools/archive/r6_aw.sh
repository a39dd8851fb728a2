#!/bin/bash
# round 6: with a segment per isoform count, is sampler_grp_all still ahead of the launches per class?
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aw; mkdir -p $O
for v in 0 1 0 1; do
  echo "== MISO_NO_PE_ALL=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_NO_PE_ALL=1; else unset MISO_NO_PE_ALL; fi
  timeout 900 python tools/r6_clock.py pe_mix --reps 4 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
