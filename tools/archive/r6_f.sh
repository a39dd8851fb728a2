#!/bin/bash
# round 6, sixth GPU call: GPU suite on the build whose units with hidden loads are assembled from kept text; every
# kernel family's row once more (did anything move?); three wavefronts per SIMD for the several-rounds two-isoform kernel.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -20
timeout 900 python tools/r6_clock.py main se_k2_hg19 se_k2_defaults se_k2_hg19_defaults pe_k2 se_k5 se_k10 se_k5_hg19 --reps 4 2>&1 | grep -E "kernels|median" > $O/rows.txt
timeout 900 python tools/r6_clock.py pe_k5 pe_k10 pe_k5_hg19 pe_mix pe_mix_hg19 --reps 3 --probe 0 2>&1 | grep -E "kernels|median" >> $O/rows.txt
cat $O/rows.txt
