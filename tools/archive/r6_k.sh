#!/bin/bash
# round 6, twelfth GPU call: the narrow several-rounds two-isoform kernel (three wavefronts per SIMD) beside the general one
# (two): GPU suite, the rows, then the rocprofv3 passes of the two-isoform rows again.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -10
for n in "" 1; do
  echo "== MISO_K2_NO_NARROW=$n" >> $O/rows.txt
  MISO_K2_NO_NARROW=$n timeout 600 python tools/r6_clock.py se_k2_defaults se_k2_hg19_defaults --reps 5 2>&1 | grep -E "kernels|median" >> $O/rows.txt
done
cat $O/rows.txt
ROUND=06 bash tools/round6_profiles.sh se_k2 se_k2_defaults pe_k2 se_k2_hg19 se_k2_hg19_defaults pe_k2_hg19 > $O/profiles.log 2>&1
tail -3 $O/profiles.log
