#!/bin/bash
# round 6, fourth GPU call: GPU suite; two-isoform rows after the NR < 3 fix; end to end with the four-stage pipeline.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -20
timeout 600 python tools/r6_clock.py main se_k2_hg19 se_k2_defaults se_k2_hg19_defaults pe_k2 pe_k2_hg19 --reps 5 2>&1 | grep -E "kernels|median" > $O/k2_rows.txt
cat $O/k2_rows.txt
MISO_TIMING=1 timeout 900 python tools/e2e_bench.py --events 40000 --reads 1000 --runs 1:fork --summary-only > $O/e2e_40000.txt 2>&1
grep -E "^miso --run|^events|Collected|main\(\)|decoded" $O/e2e_40000.txt
