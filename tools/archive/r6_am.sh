#!/bin/bash
# round 6: sampler_flat, five to eight isoforms: the chains per wavefront sized for two or for three workgroups per CU (runtime.hip flat_wgs_for)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6am; mkdir -p $O
for w in 2 3 2 3; do
  echo "== MISO_FLAT_WGS=$w" >> $O/ab.txt
  MISO_FLAT_WGS=$w timeout 900 python tools/archive/r6_shape.py K=5 K=6 K=7 K=8 K=5,reads=hg19 K=6,reads=hg19 K=7,reads=hg19 K=8,reads=hg19 K=13 K=16 K=13,reads=hg19 K=16,reads=hg19 --reps 2 2>&1 | grep median >> $O/ab.txt
done
for f in 0.8 0.85 0.9 0.95; do
  echo "== MISO_FLAT_WGS=3 MISO_FLAT_ROUNDS_FRAC=$f" >> $O/ab.txt
  MISO_FLAT_WGS=3 MISO_FLAT_ROUNDS_FRAC=$f timeout 900 python tools/archive/r6_shape.py K=5,reads=hg19 K=8,reads=hg19 --reps 2 2>&1 | grep median >> $O/ab.txt
done
cat $O/ab.txt
