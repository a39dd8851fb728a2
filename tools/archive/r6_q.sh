#!/bin/bash
# round 6: whole-gene mixes with every isoform-count class on CUs of its own (MISO_CLASS_CUMASK=1, runtime.hip)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
MISO_CLASS_CUMASK=1 timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_fuzz.py tests/test_gpu_paired_dense.py tests/test_gpu_heavy_tail.py 2>&1 | tail -3
for c in 0 1 0 1; do
  echo "== MISO_CLASS_CUMASK=$c" >> $O/cumask.txt
  MISO_TIMING=1 MISO_CLASS_CUMASK=$c timeout 600 python tools/r6_clock.py pe_mix pe_mix_hg19 --reps 5 --probe 0 2>&1 | grep -E "median|class [0-9]+: CUs|gave up" | sort | uniq -c | sort -k2 >> $O/cumask.txt
done
cat $O/cumask.txt
