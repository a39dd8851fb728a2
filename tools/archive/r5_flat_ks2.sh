#!/bin/bash
# round 5: sampler_flat<KC, KS> (the slice layout of the launch's largest isoform count at compile time) in the library:
# the whole GPU suite, then rows against MISO_FLAT_NO_KS=1
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_ks2.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/flat_ks2_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/flat_ks2_tests.log | tail -1)" >> $out
MISO_FLAT_NO_KS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r5/flat_ks2_tests_noks.log 2>&1
echo "run-time layout (MISO_FLAT_NO_KS=1): tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5/flat_ks2_tests_noks.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 3" "--K 4" "--K 5" "--K 6" "--K 7" "--K 8" "--K 9" "--K 10" "--K 12" "--K 5 --reads-dist hg19" "--K 10 --reads-dist hg19" "--K-range 3 20 --events 16384" "--K 5 --chains 6 --iters 5000 --burn 500 --lag 10 --events 20000"; do
  run "compile-time $cfg" $cfg -- MISO_X=0
  run "run-time     $cfg" $cfg -- MISO_FLAT_NO_KS=1
done
cat $out
