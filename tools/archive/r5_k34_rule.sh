#!/bin/bash
# round 5: the eight-lane rule extended to three and four isoforms (score tables in global memory): parity and rows
mkdir -p gpurun_out/r5
out=gpurun_out/r5/k34_rule.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/k34_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/k34_tests.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:60], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 3 --paired" "--K 4 --paired" "--K 3 --paired --reads-dist hg19" "--K 4 --paired --reads-dist hg19" "--K-range 3 4 --paired" "--K 3 --paired --chains 6 --iters 5000 --burn 500 --lag 10 --events 20000" "--K 3 --paired --events 10000"; do
  run "rule $cfg" $cfg -- MISO_X=0
  run "16   $cfg" $cfg -- MISO_PE_LANES8=0
done
cat $out
