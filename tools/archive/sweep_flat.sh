#!/bin/bash
# flat vs grouped single-end kernel across isoform counts (full iterations); gpurun_out/r02/sweep_flat.txt
mkdir -p gpurun_out/r02
out=gpurun_out/r02/sweep_flat.txt; : > $out
for K in 3 4 5 6 8 10 12 16 20; do
  E=40000; [ $K -ge 16 ] && E=8192
  for mode in flat grp; do
    if [ $mode = grp ]; then export MISO_NO_FLAT=1; else unset MISO_NO_FLAT; fi
    python bench.py --no-cpu-baseline --no-matrix --K $K --events $E --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('K=$K events=$E $mode', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
  done
done
cat $out
