#!/bin/bash
# round 6: small genes of a whole-gene paired-end batch: sixteen or eight lanes per chain?  (16 384 genes of 3 - 20 isoforms, 100 / 250 / 500 pairs each)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ay; mkdir -p $O
S="K=3-20,paired=1,reads=100 K=3-20,paired=1,reads=250 K=3-20,paired=1,reads=500"
echo "== default" >> $O/ab.txt
timeout 900 python tools/archive/r6_shape.py $S --events 16384 --reps 2 2>&1 | grep median >> $O/ab.txt
echo "== MISO_PE_MULTI=1 MISO_GENERAL_LANES_BY_CLASS 8 lanes (multi off by the env), buckets off" >> $O/ab.txt
MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:8,12:8,16:8,32:8 timeout 900 python tools/archive/r6_shape.py $S --events 16384 --reps 2 2>&1 | grep median >> $O/ab.txt
echo "== 16 lanes, buckets off, per-run launches" >> $O/ab.txt
MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:16 timeout 900 python tools/archive/r6_shape.py $S --events 16384 --reps 2 2>&1 | grep median >> $O/ab.txt
echo "== 4:8,8:8 only" >> $O/ab.txt
MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:8,12:16,16:16,32:16 timeout 900 python tools/archive/r6_shape.py $S --events 16384 --reps 2 2>&1 | grep median >> $O/ab.txt
cat $O/ab.txt
