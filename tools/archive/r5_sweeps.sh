#!/bin/bash
# round 5, after the prefetch / gather fixes: chains per wavefront of sampler_flat and lanes per chain of the paired-end kernels again
mkdir -p gpurun_out/r5
out=gpurun_out/r5/sweeps.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:70], d['roofline']['kernel_ms'])" >> $out
}
for K in 5 6 8; do
  run "SE K=$K auto" --K $K -- MISO_X=0
  for nc in 6 7 9 10 12; do run "SE K=$K" --K $K -- MISO_FLAT_NC=$nc; done
done
run "SE K=10 auto" --K 10 -- MISO_X=0
for nc in 4 6 7; do run "SE K=10" --K 10 -- MISO_FLAT_NC=$nc; done
run "SE K=4 auto" --K 4 -- MISO_X=0
for nc in 8 10 12 16; do run "SE K=4" --K 4 -- MISO_FLAT_NC=$nc; done
for K in 5 8; do
  run "PE K=$K auto" --K $K --paired -- MISO_X=0
  run "PE K=$K 8 lanes" --K $K --paired -- MISO_GENERAL_LANES=8
  run "PE K=$K 32 lanes" --K $K --paired -- MISO_GENERAL_LANES=32
done
run "PE K=10 auto" --K 10 --paired --events 20000 -- MISO_X=0
run "PE K=10 32 lanes" --K 10 --paired --events 20000 -- MISO_GENERAL_LANES=32
run "PE K=10 8 lanes" --K 10 --paired --events 20000 -- MISO_GENERAL_LANES=8
cat $out
