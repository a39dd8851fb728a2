#!/bin/bash
# round 5: sampler_flat's wavefronts packed by units + the per-chain scalar step (30 K + 25 unit equivalents) against by
# units alone (MISO_FLAT_PACK_OV=0), hg19-like read counts
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_pack_ov.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_heavy_tail.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/flat_pack_ov_tests.log 2>&1
echo "tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/flat_pack_ov_tests.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:50], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 3 --reads-dist hg19" "--K 5 --reads-dist hg19" "--K 8 --reads-dist hg19" "--K 10 --reads-dist hg19" "--K 16 --reads-dist hg19 --events 20000" "--K-range 3 20 --events 16384 --reads-dist hg19" "--K 5 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 --events 20000" "--K 5"; do
  run "cost  $cfg" $cfg -- MISO_X=0
  run "units $cfg" $cfg -- MISO_FLAT_PACK_OV=0
done
cat $out
