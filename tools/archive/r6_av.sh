#!/bin/bash
# round 6: no wavefront with chains of two isoform counts -- a segment per isoform count in sampler_grp_multi / sampler_grp_all (MISO_PE_NO_KSPLIT=1 MISO_PE_ALL_ORDER=0: before)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6av; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
grep -v "no chains" $O/tests.log | grep -E "^E  |FAILED" | head -10
for v in 0 1 0 1; do
  echo "== before=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_PE_NO_KSPLIT=1 MISO_PE_ALL_ORDER=0; else unset MISO_PE_NO_KSPLIT MISO_PE_ALL_ORDER; fi
  timeout 900 python tools/r6_clock.py pe_mix pe_mix_hg19 --reps 4 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
unset MISO_PE_NO_KSPLIT MISO_PE_ALL_ORDER
timeout 600 python tools/archive/r6_shape.py K=5,paired=1,reads=hg19 K=10,paired=1,reads=hg19 --reps 2 2>&1 | grep median >> $O/ab.txt
cat $O/ab.txt
