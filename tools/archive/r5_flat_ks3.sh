#!/bin/bash
# round 5: sampler_flat<KC, KS> for 13 - 20 isoforms: GPU suite, rows against MISO_FLAT_NO_KS=1
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_ks3.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/flat_ks3_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/flat_ks3_tests.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:60], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 13 --events 20000" "--K 14 --events 20000" "--K 16 --events 20000" "--K 17 --events 20000" "--K 20 --events 20000" "--K-range 3 20 --events 16384" "--K-range 3 20 --events 16384 --reads-dist hg19"; do
  run "compile-time $cfg" $cfg -- MISO_X=0
  run "run-time     $cfg" $cfg -- MISO_FLAT_NO_KS=1
done
cat $out
