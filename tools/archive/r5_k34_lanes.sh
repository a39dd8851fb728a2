#!/bin/bash
# round 5: paired-end, three and four isoforms: lanes per chain when the score table stays in global memory (the LDS copy
# holds the class at four chains per wavefront)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/k34_lanes.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
for K in 3 4; do
  run "PE K=$K rule" --K $K --paired -- MISO_X=0
  for kb in 60 40; do
    for g in 16 8 4; do run "PE K=$K" --K $K --paired -- MISO_LDS_MAX_KB=$kb MISO_GENERAL_LANES=$g; done
  done
done
run "PE K=3 hg19 rule" --K 3 --paired --reads-dist hg19 -- MISO_X=0
run "PE K=3 hg19" --K 3 --paired --reads-dist hg19 -- MISO_LDS_MAX_KB=40 MISO_GENERAL_LANES=8
cat $out
