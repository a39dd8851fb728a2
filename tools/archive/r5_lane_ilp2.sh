#!/bin/bash
# round 5: sampler_lane_ilp with the exp / log coefficients in registers; parity, then A/B against the lean form
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lane_ilp2.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_collapsed.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r5/lane_ilp2_tests.log 2>&1
echo "collapsed + fuzz tests rc=$? $(tail -1 gpurun_out/r5/lane_ilp2_tests.log)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/lane_ilp2_tests.log | head -10 >> $out
python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  run "ilp" --collapsed 1 -- MISO_X=0
  run "lean" --collapsed 1 -- MISO_LANE_ILP=0
done
run "ilp reads=3000" --collapsed 1 --reads 3000 -- MISO_X=0
run "lean reads=3000" --collapsed 1 --reads 3000 -- MISO_LANE_ILP=0
run "hg19 ilp" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "hg19 lean" --collapsed 1 --reads-dist hg19 -- MISO_LANE_ILP=0
run "defaults auto" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "defaults ilp forced" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_LANE_ILP=1
run "131072 events ilp" --collapsed 1 --events 131072 -- MISO_LANE_ILP=1
run "131072 events lean" --collapsed 1 --events 131072 -- MISO_LANE_ILP=0
run "mix hg19" --K-range 3 20 --paired --events 16384 --reads-dist hg19 -- MISO_X=0
run "mix" --K-range 3 20 --paired --events 16384 -- MISO_X=0
cat $out
