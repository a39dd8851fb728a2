#!/bin/bash
# round 5: paired-end lanes per chain again (8 lanes won at five isoforms after the gather / order changes)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/sweeps2.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:70], d['roofline']['kernel_ms'])" >> $out
}
for K in 3 4 5 6 7 8 9; do
  for g in 4 8 16; do run "PE K=$K" --K $K --paired -- MISO_GENERAL_LANES=$g; done
done
run "PE K=12 8 lanes" --K 12 --paired --events 20000 -- MISO_GENERAL_LANES=8
run "PE K=12 16 lanes" --K 12 --paired --events 20000 -- MISO_GENERAL_LANES=16
M="--K-range 3 20 --paired --events 16384"
run "mix auto" $M -- MISO_X=0
run "mix 8 up to 8" $M -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:8,12:16,16:16,32:16
run "mix 8 up to 12" $M -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:8,8:8,12:8,16:16,32:16
run "K=5 hg19 auto" --K 5 --paired --reads-dist hg19 -- MISO_X=0
cat $out
