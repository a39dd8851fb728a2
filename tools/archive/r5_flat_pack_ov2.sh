#!/bin/bash
# round 5: the final packing rule (units + 16 K + 14 per chain, round overshoot corrected) against by units alone (MISO_FLAT_PACK_OV=0)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_pack_ov2.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_heavy_tail.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q > gpurun_out/r5/flat_pack_ov2_tests.log 2>&1
echo "tests rc=$? $(grep -E "passed|failed" gpurun_out/r5/flat_pack_ov2_tests.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:30], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 3 --reads-dist hg19" "--K 4 --reads-dist hg19" "--K 5 --reads-dist hg19" "--K 6 --reads-dist hg19" "--K 8 --reads-dist hg19" "--K 10 --reads-dist hg19" "--K 16 --reads-dist hg19 --events 20000" "--K-range 3 20 --events 16384 --reads-dist hg19" "--K 5 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 --events 20000"; do
  for pct in 100 0; do run "$cfg" $cfg -- MISO_FLAT_PACK_OV=$pct; done
done
cat $out
