#!/bin/bash
# round 5: sampler_flat with the slice layout of ONE isoform count as compile-time constants (variant libraries, valid for
# that count only) against the run-time layout
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_ks.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
MISO_AMD_LIB=tools/_build/libmiso_flatks5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "single_end" > gpurun_out/r5/flat_ks_tests.log 2>&1
echo "KS=5 library, single-end parity tests (other counts than five MAY fail) rc=$? $(grep -E 'passed|failed' gpurun_out/r5/flat_ks_tests.log | tail -1)" >> $out
for rep in 1 2; do
  run "run-time ks K=5" --K 5 -- MISO_X=0
  run "fixed ks   K=5" --K 5 -- MISO_AMD_LIB=tools/_build/libmiso_flatks5.so
  run "run-time ks K=10" --K 10 -- MISO_X=0
  run "fixed ks   K=10" --K 10 -- MISO_AMD_LIB=tools/_build/libmiso_flatks10.so
done
run "run-time ks K=5 hg19" --K 5 --reads-dist hg19 -- MISO_X=0
run "fixed ks   K=5 hg19" --K 5 --reads-dist hg19 -- MISO_AMD_LIB=tools/_build/libmiso_flatks5.so
cat $out
