#!/bin/bash
# round 5: pe_dense without the per-trip spill of the block's Philox words (the cold path draws them again); on top of it the
# late gather for EVERY isoform count (variant library) against up to six + asm records for seven and eight (in-tree)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/ab4.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "paired or pe or golden or heavy or fuzz or frontend or parity" > gpurun_out/r5/ab4_tests.log 2>&1
echo "gpu tests (in-tree) rc=$? $(grep -E 'passed|failed' gpurun_out/r5/ab4_tests.log | tail -1)" >> $out
V=tools/_build/libmiso_late20.so
MISO_AMD_LIB=$V timeout 1500 python -m pytest tests -m gpu -x -q -k "paired or pe or golden or heavy or fuzz or parity" > gpurun_out/r5/ab4_tests_v.log 2>&1
echo "gpu tests (late gather everywhere) rc=$? $(grep -E 'passed|failed' gpurun_out/r5/ab4_tests_v.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:70], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 5 --paired" "--K 7 --paired" "--K 8 --paired" "--K 10 --paired --events 20000" "--K 12 --paired --events 20000" "--K 16 --paired --events 20000" "--K 20 --paired --events 20000" \
           "--K-range 3 20 --paired --events 16384" "--K-range 3 20 --paired --events 16384 --reads-dist hg19"; do
  run "in-tree  $cfg" $cfg -- MISO_X=0
  run "late all $cfg" $cfg -- MISO_AMD_LIB=$V
done
cat $out
