#!/bin/bash
# round 5: eight lanes per chain for single-class paired-end batches of five to nine isoforms: parity, rows, and the same
# for the size buckets' normal run / in a mix (MISO_PE_LANES8=1)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lanes8.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/lanes8_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/lanes8_tests.log | tail -1)" >> $out
MISO_PE_LANES8=1 timeout 1500 python -m pytest tests -m gpu -x -q -k "paired or pe or golden or heavy or fuzz or parity" > gpurun_out/r5/lanes8_tests_forced.log 2>&1
echo "gpu tests with MISO_PE_LANES8=1 rc=$? $(grep -E 'passed|failed' gpurun_out/r5/lanes8_tests_forced.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:70], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 5 --paired" "--K 7 --paired" "--K 9 --paired" "--K 10 --paired --events 20000" "--K 5 --paired --events 10000" "--K 5 --paired --events 20000"; do
  run "auto $cfg" $cfg -- MISO_X=0
  run "16   $cfg" $cfg -- MISO_PE_LANES8=0
done
for cfg in "--K 5 --paired --reads-dist hg19" "--K 8 --paired --reads-dist hg19" "--K-range 3 20 --paired --events 16384" "--K-range 3 20 --paired --events 16384 --reads-dist hg19" "--K 5 --paired --chains 6 --iters 5000 --burn 500 --lag 10 --events 20000"; do
  run "auto   $cfg" $cfg -- MISO_X=0
  run "lanes8 $cfg" $cfg -- MISO_PE_LANES8=1
done
cat $out
