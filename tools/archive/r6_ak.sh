#!/bin/bash
# round 6: 24-bit multiplies in sampler_flat's slice addressing; the narrow two-isoform kernel without machine-level hoisting
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ak; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "flat or parity or fuzz or heavy or wide or convergent or isoform" > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
timeout 900 python tools/r6_clock.py se_k5 se_k10 se_k5_hg19 se_k2_defaults --reps 3 2>&1 | grep -E "kernels|median" | cut -c1-150 > $O/rows.txt
cat $O/rows.txt
