#!/bin/bash
# round 6, second GPU call: the GPU suite on the new build (CONVERGENT_MEAN window, RoundOpen in every kernel), the
# SIMD partners' priority balance A/B (MISO_K2_BALANCE=0/1, same box), the end-to-end run's stage breakdown.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -10
for b in 1 0 1 0; do
  echo "== MISO_K2_BALANCE=$b" >> $O/balance.txt
  MISO_K2_BALANCE=$b timeout 300 python tools/r6_clock.py main se_k2_hg19 --reps 6 2>&1 | grep -E "kernels|median" >> $O/balance.txt
done
cat $O/balance.txt
MISO_TIMING=1 timeout 900 python tools/e2e_bench.py --events 40000 --reads 1000 --runs 1:fork --summary-only > $O/e2e_40000.txt 2>&1
cat $O/e2e_40000.txt | tail -40
