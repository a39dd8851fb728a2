#!/bin/bash
# round 5: pe_dense's records through loads the compiler does not see (seven isoforms and more), the collapsed slot list by
# binomial regime; parity first
mkdir -p gpurun_out/r5
out=gpurun_out/r5/ab3.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/ab3_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/ab3_tests.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/ab3_tests.log | head -10 >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 7 --paired" "--K 8 --paired" "--K 10 --paired --events 20000" "--K 12 --paired --events 20000" "--K 16 --paired --events 20000" "--K 20 --paired --events 20000" \
           "--K-range 3 20 --paired --events 16384" "--K-range 3 20 --paired --events 16384 --reads-dist hg19" "--K 5 --paired" "--K 10 --paired --events 20000 --reads-dist hg19"; do
  run "now $cfg" $cfg -- MISO_X=0
done
for rep in 1 2; do
  run "collapsed regime order" --collapsed 1 -- MISO_X=0
  run "collapsed by reads only" --collapsed 1 -- MISO_LANE_NO_REGIME_ORDER=1
done
run "collapsed hg19 regime order" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "collapsed hg19 by reads only" --collapsed 1 --reads-dist hg19 -- MISO_LANE_NO_REGIME_ORDER=1
run "collapsed defaults regime order" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "collapsed defaults by reads only" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_LANE_NO_REGIME_ORDER=1
run "defaults" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "defaults" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "headline" --K 2 -- MISO_X=0
cat $out
