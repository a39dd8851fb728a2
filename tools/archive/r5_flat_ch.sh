#!/bin/bash
# round 5: sampler_flat's leader sums in chunks of four isoforms instead of eight (registers: 209 -> 183 unconstrained,
# scratch 176 -> 48 bytes at the 168 of three workgroups per CU)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_ch.txt; : > $out
V=tools/_build/libmiso_flatch4.so
MISO_AMD_LIB=$V timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_heavy_tail.py -x -q > gpurun_out/r5/flat_ch_tests.log 2>&1
echo "chunks of four: tests rc=$? $(tail -1 gpurun_out/r5/flat_ch_tests.log)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 5" "--K 6" "--K 7" "--K 8" "--K 10" "--K 12" "--K 14 --events 20000" "--K 16 --events 20000" "--K 5 --reads-dist hg19" "--K 10 --reads-dist hg19" "--K-range 3 20 --events 16384"; do
  run "eight $cfg" $cfg -- MISO_X=0
  run "four  $cfg" $cfg -- MISO_AMD_LIB=$V
done
cat $out
