#!/bin/bash
# round 6: the small genes of a whole-gene paired-end batch on eight lanes per chain (bucket -1, MISO_PE_T_SMALL quads; 0 = off)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ba; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
grep -v "no chains" $O/tests.log | grep -E "^E  |FAILED" | head -10
for t in 0 96 48 160 0 96; do
  echo "== MISO_PE_T_SMALL=$t" >> $O/ab.txt
  MISO_PE_T_SMALL=$t timeout 900 python tools/r6_clock.py pe_mix_hg19 --reps 3 --probe 0 2>&1 | grep -E "median" | cut -c1-150 >> $O/ab.txt
done
MISO_PE_T_SMALL=96 timeout 900 python tools/archive/r6_shape.py K=3-20,paired=1,reads=100 K=3-20,paired=1,reads=250 --events 16384 --reps 2 2>&1 | grep median | cut -c1-200 >> $O/ab.txt
MISO_PE_T_SMALL=0 timeout 900 python tools/archive/r6_shape.py K=3-20,paired=1,reads=100 K=3-20,paired=1,reads=250 --events 16384 --reps 2 2>&1 | grep median | cut -c1-200 >> $O/ab.txt
cat $O/ab.txt
