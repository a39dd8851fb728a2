#!/bin/bash
# round 6: bucketed whole-gene batches: the sixteen-lane runs of every class in sampler_grp_all (pieces ordered by cost), the heavier kinds per class (MISO_PE_ALL16_HYBRID=1)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6at; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whole_gene or heavy or scale" > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
MISO_PE_ALL16_HYBRID=1 timeout 900 python -m pytest tests -m gpu -x -q -k "heavy or fuzz" > $O/tests_h.log 2>&1; echo "hybrid tests rc=$? $(grep -v 'no chains' $O/tests_h.log | tail -1)"
for v in 0 1 0 1; do
  echo "== MISO_PE_ALL16_HYBRID=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_PE_ALL16_HYBRID=1; else unset MISO_PE_ALL16_HYBRID; fi
  timeout 900 python tools/r6_clock.py pe_mix_hg19 --reps 4 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
