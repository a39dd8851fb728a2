#!/bin/bash
# round 6: exp(x - x) as a constant for every lanes-per-chain layout (one lane per chain: three exponentials instead of four)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
timeout 600 python tools/r6_clock.py se_k2_defaults se_k2_hg19_defaults se_k2_hg19 main --reps 5 --probe 0 2>&1 | grep -E "kernels|median" > $O/rows.txt
cat $O/rows.txt
ROUND=06 bash tools/round6_profiles.sh se_k2_defaults se_k2_hg19_defaults se_k2_hg19 > $O/profiles.log 2>&1
tail -2 $O/profiles.log | cut -c1-200
