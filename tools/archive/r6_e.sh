#!/bin/bash
# round 6, fifth GPU call: GPU suite; two-isoform rows (old MH flow kept for one / two lanes per chain); paired-end K >= 3
# with three workgroups per CU (variant library); end to end (decode beside the annotation work).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -20
timeout 600 python tools/r6_clock.py main se_k2_hg19 se_k2_defaults se_k2_hg19_defaults pe_k2 --reps 5 2>&1 | grep -E "kernels|median" > $O/k2_rows.txt
cat $O/k2_rows.txt
for lib in "" tools/_build/libmiso_peb3.so; do
  echo "== MISO_AMD_LIB=$lib" >> $O/pe_blocks.txt
  MISO_AMD_LIB=$lib timeout 600 python tools/r6_clock.py pe_k5 pe_k10 pe_k5_hg19 --reps 3 --probe 0 2>&1 | grep -E "kernels|median" >> $O/pe_blocks.txt
done
cat $O/pe_blocks.txt
MISO_TIMING=1 timeout 900 python tools/e2e_bench.py --events 40000 --reads 1000 --runs 1:fork --summary-only > $O/e2e_40000.txt 2>&1
grep -E "^miso --run|^events|Collected|main\(\)|decoded|alignment file open" $O/e2e_40000.txt
