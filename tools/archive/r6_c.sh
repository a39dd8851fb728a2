#!/bin/bash
# round 6, third GPU call: GPU suite on the build with the merged exponential calls of the two-isoform MH step, RoundOpen
# without scratch, per-run collapsed routing; then the two-isoform rows, balance on / off.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -10
for b in 1 0; do
  echo "== MISO_K2_BALANCE=$b" >> $O/k2_rows.txt
  MISO_K2_BALANCE=$b timeout 600 python tools/r6_clock.py main se_k2_hg19 se_k2_defaults se_k2_hg19_defaults pe_k2 pe_k2_hg19 --reps 5 2>&1 | grep -E "kernels|median" >> $O/k2_rows.txt
done
cat $O/k2_rows.txt
