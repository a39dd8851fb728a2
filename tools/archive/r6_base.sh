#!/bin/bash
# round 6, first GPU call: the clock probe (does it see the launch? does it cost anything?), kernel time x clock of the
# headline and of `se_k2_defaults` over ten launches, rocm-smi's view beside it, then the GPU suite.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
O=gpurun_out/r6a
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.txt 2>&1 &
SMI=$!
timeout 600 python tools/r6_clock.py main se_k2_defaults se_k2_hg19_defaults pe_k5 se_k5 --reps 10 > $O/clock_probe.txt 2>&1
timeout 300 python tools/r6_clock.py main se_k2_defaults --reps 10 --probe 0 > $O/clock_noprobe.txt 2>&1
kill $SMI
sort $O/smi.txt | uniq -c | sort -rn | head -8 > $O/smi_hist.txt
cat $O/clock_probe.txt | grep -E "kernels|median"; cat $O/clock_noprobe.txt | grep -E "kernels|median"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -10
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.out 2> $O/bench.err
echo "bench rc=$?"; tail -c 3900 $O/bench.out
