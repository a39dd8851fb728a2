#!/bin/bash
# round 6: whole-gene batches with size buckets as one launch per KIND of run (sampler_grp_all<LANES>) against a launch per class
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6as; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whole_gene or heavy or paired or fuzz" > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
grep -v "no chains" $O/tests.log | grep -E "^E |FAILED" | head -10
for v in 0 1 0 1; do
  echo "== MISO_NO_PE_ALL=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_NO_PE_ALL=1; else unset MISO_NO_PE_ALL; fi
  timeout 900 python tools/r6_clock.py pe_mix pe_mix_hg19 --reps 4 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
