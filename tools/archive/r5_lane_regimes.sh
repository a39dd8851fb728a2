#!/bin/bash
# round 5: what the slowest wavefront of sampler_lane is slow at: reads per event (the binomial's regime), no binomial, no scores
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lane_regimes.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for r in 100 300 1000 3000 10000; do
  run "ILP reads=$r" --collapsed 1 --reads $r -- MISO_X=0
done
run "old reads=3000" --collapsed 1 --reads 3000 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
run "old reads=300" --collapsed 1 --reads 300 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
run "no binomial (old form)" --collapsed 1 -- MISO_AMD_LIB=tools/_build/libmiso_lane_nobinom.so
run "no scores (old form)" --collapsed 1 -- MISO_AMD_LIB=tools/_build/libmiso_lane_nomh.so
cat $out
