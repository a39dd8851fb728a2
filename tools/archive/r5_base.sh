#!/bin/bash
# round 5: the AS-class fixtures through the device path + this box's numbers for the rows the round works on;
# gpurun_out/r5/base.txt
mkdir -p gpurun_out/r5
out=gpurun_out/r5/base.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_golden.py -m gpu -x -q -k "device_match or golden_inputs" > gpurun_out/r5/as_tests.log 2>&1
echo "AS-class gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/as_tests.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/as_tests.log | head -20 >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
run "headline" --K 2 -- MISO_X=0
run "collapsed" --K 2 --collapsed 1 -- MISO_X=0
for K in 3 5 10; do run "SE K=$K" --K $K -- MISO_X=0; done
run "PE K=3" --K 3 --paired -- MISO_X=0
run "PE K=5" --K 5 --paired -- MISO_X=0
run "PE K=10" --K 10 --paired --events 20000 -- MISO_X=0
run "PE mix" --K-range 3 20 --paired --events 16384 -- MISO_X=0
run "PE mix hg19" --K-range 3 20 --paired --events 16384 --reads-dist hg19 -- MISO_X=0
cat $out
