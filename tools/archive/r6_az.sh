#!/bin/bash
# round 6: small genes, classes 4 and 8 on eight lanes per chain (MISO_PE_LANES8=1) against sixteen
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6az; mkdir -p $O
S="K=3-8,paired=1,reads=100 K=3-8,paired=1,reads=250 K=3-8,paired=1,reads=500 K=3-20,paired=1,reads=100"
for v in 0 1; do
  echo "== MISO_PE_LANES8=$v (MISO_NO_PE_ALL=1)" >> $O/ab.txt
  MISO_NO_PE_ALL=1 MISO_PE_LANES8=$v timeout 900 python tools/archive/r6_shape.py $S --events 16384 --reps 2 2>&1 | grep median | cut -c1-200 >> $O/ab.txt
done
cat $O/ab.txt
