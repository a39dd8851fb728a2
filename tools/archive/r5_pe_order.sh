#!/bin/bash
# round 5: paired-end draws in the order of their fragment-length rows: GPU parity, then the paired-end rows
mkdir -p gpurun_out/r5
out=gpurun_out/r5/pe_order.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "paired or pe or golden or heavy or fuzz or match or frontend" > gpurun_out/r5/pe_order_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/pe_order_tests.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/pe_order_tests.log | head -20 >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
run "PE K=2" --K 2 --paired -- MISO_X=0
run "PE K=2 hg19" --K 2 --paired --reads-dist hg19 -- MISO_X=0
run "PE K=3" --K 3 --paired -- MISO_X=0
run "PE K=5" --K 5 --paired -- MISO_X=0
run "PE K=5 hg19" --K 5 --paired --reads-dist hg19 -- MISO_X=0
run "PE K=8" --K 8 --paired -- MISO_X=0
run "PE K=10" --K 10 --paired --events 20000 -- MISO_X=0
run "PE K=16" --K 16 --paired --events 20000 -- MISO_X=0
run "PE mix" --K-range 3 20 --paired --events 16384 -- MISO_X=0
run "PE mix hg19" --K-range 3 20 --paired --events 16384 --reads-dist hg19 -- MISO_X=0
cat $out
