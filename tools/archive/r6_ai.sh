#!/bin/bash
# round 6: the whole GPU suite and every row's kernel time on the build with sampler_flat<KC, KS, UNI> and the no-hoisting flag on flat + grp units
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ai; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
timeout 1200 python tools/r6_clock.py se_k5 se_k10 se_k5_hg19 pe_k5 pe_k10 pe_k5_hg19 pe_mix pe_mix_hg19 --reps 3 2>&1 | grep -E "kernels|median" | cut -c1-150 > $O/rows.txt
cat $O/rows.txt
