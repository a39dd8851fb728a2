#!/bin/bash
# round 5: paired-end two-isoform events, lanes per chain forced (single-width launches) against the planner's choice
mkdir -p gpurun_out/r5
out=gpurun_out/r5/pek2_lanes.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'], 'lanes', d.get('lanes_per_chain'))" >> $out
}
for cfg in "--K 2 --paired" "--K 2 --paired --reads-dist hg19" "--K 2" "--K 2 --chains 6 --iters 5000 --burn 500 --lag 10"; do
  run "planner $cfg" $cfg -- MISO_X=0
  for g in 2 4 8 16 32; do run "forced  $cfg" $cfg -- MISO_LANES_PER_CHAIN=$g; done
done
cat $out
