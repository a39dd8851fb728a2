#!/bin/bash
# round 5: collapsed mode, chains per wavefront when the batch under-fills the device: 1, 2 or 3 wavefronts per SIMD
mkdir -p gpurun_out/r5
out=gpurun_out/r5/spread2.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
MISO_LANE_SPREAD=2 timeout 600 python -m pytest tests/test_gpu_collapsed.py -x -q > gpurun_out/r5/spread2_tests.log 2>&1
echo "collapsed tests, two per SIMD rc=$? $(tail -1 gpurun_out/r5/spread2_tests.log)" >> $out
for rep in 1 2; do
  for n in 1 2 3 4 0; do run "headline per_simd=$n" --collapsed 1 -- MISO_LANE_SPREAD=$n; done
done
for n in 1 2 3 0; do run "hg19 per_simd=$n" --collapsed 1 --reads-dist hg19 -- MISO_LANE_SPREAD=$n; done
for n in 1 2 3 0; do run "reads=3000 per_simd=$n" --collapsed 1 --reads 3000 -- MISO_LANE_SPREAD=$n; done
for n in 1 2 3 0; do run "20000 events per_simd=$n" --collapsed 1 --events 20000 -- MISO_LANE_SPREAD=$n; done
for n in 1 2 0; do run "65536 events per_simd=$n" --collapsed 1 --events 65536 -- MISO_LANE_SPREAD=$n; done
cat $out
