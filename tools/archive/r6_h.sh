#!/bin/bash
# round 6, eighth GPU call: (a) is the paired-end K >= 3 read loop waiting for its records?  same chains over fewer events
# (tools/r6_pe_ws.py); (b) whole-gene mixes with the wavefronts' issue priority by class (variant library, MISO_CLASS_PRIO).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
timeout 900 python tools/r6_pe_ws.py 5 10 > $O/pe_working_set.txt 2>&1
cat $O/pe_working_set.txt
export MISO_AMD_LIB=tools/_build/libmiso_clsprio.so
timeout 300 python -m pytest tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -2
for c in 0 1 2 0 1 2; do
  echo "== MISO_CLASS_PRIO=$c" >> $O/class_prio.txt
  MISO_CLASS_PRIO=$c timeout 600 python tools/r6_clock.py pe_mix pe_mix_hg19 --reps 5 --probe 0 2>&1 | grep -E "median" >> $O/class_prio.txt
done
cat $O/class_prio.txt
