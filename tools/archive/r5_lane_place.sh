#!/bin/bash
# round 5: one workgroup of sampler_lane per CU (an LDS allocation that leaves no room for a second) against the dispatcher's placement
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lane_place.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  for pad in 84 0 56 40; do
    run "ILP pad=$pad" --collapsed 1 -- MISO_LANE_LDS_PAD_KB=$pad
    run "old pad=$pad" --collapsed 1 -- MISO_LANE_LDS_PAD_KB=$pad MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
  done
done
run "hg19 ILP auto" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "hg19 ILP pad=0" --collapsed 1 --reads-dist hg19 -- MISO_LANE_LDS_PAD_KB=0
run "20000 events ILP auto" --collapsed 1 --events 20000 -- MISO_X=0
run "20000 events ILP pad=0" --collapsed 1 --events 20000 -- MISO_LANE_LDS_PAD_KB=0
run "65536 events ILP auto" --collapsed 1 --events 65536 -- MISO_X=0
run "65536 events ILP pad=0" --collapsed 1 --events 65536 -- MISO_LANE_LDS_PAD_KB=0
MISO_LANE_LDS_PAD_KB=84 timeout 600 python -m pytest tests/test_gpu_collapsed.py -x -q > gpurun_out/r5/lane_place_tests.log 2>&1
echo "collapsed tests with the pad rc=$? $(tail -1 gpurun_out/r5/lane_place_tests.log)" >> $out
cat $out
