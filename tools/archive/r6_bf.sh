#!/bin/bash
# round 6: a single paired-end class on eight lanes per chain with several isoform counts through sampler_grp_multi (a segment per count) against its own launch (MISO_NO_PE_MULTI=1)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bf; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
grep -v "no chains" $O/tests.log | grep -E "^E  |FAILED" | head -10
for v in 0 1 0 1; do
  echo "== MISO_NO_PE_MULTI=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_NO_PE_MULTI=1; else unset MISO_NO_PE_MULTI; fi
  timeout 900 python tools/archive/r6_shape.py K=5-8,paired=1 K=5-8,paired=1,reads=250 K=3-4,paired=1 --events 16384 --reps 2 2>&1 | grep median | cut -c1-160 >> $O/ab.txt
done
cat $O/ab.txt
