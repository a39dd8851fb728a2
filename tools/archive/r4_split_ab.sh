#!/bin/bash
# round 4: lazy low bits (include/miso_philox.h) -- GPU parity of the two-isoform single-end kernels, then the rows they
# serve, product build against variants of the read loop's blocks in flight (MISO_K2_UQ); gpurun_out/r4/split_ab.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/split_ab.txt; : > $out
MISO_AMD_LIB=${TEST_LIB:-miso_amd/libmiso_amd.so} timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_heavy_tail.py -m gpu -x -q > gpurun_out/r4/gputests_split.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r4/gputests_split.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r4/gputests_split.log | head -20 >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for e in MISO_X=0 MISO_AMD_LIB=tools/_build/libmiso_swar.so MISO_AMD_LIB=tools/_build/libmiso_swaruq4.so MISO_AMD_LIB=tools/_build/libmiso_nosdwa.so MISO_AMD_LIB=tools/_build/libmiso_uq4.so; do
  for rep in 1 2; do
    run "headline" --K 2 -- $e
  done
  run "defaults" --K 2 --chains 6 --iters 5000 --burn 500 --lag 10 -- $e
  run "hg19" --K 2 --reads-dist hg19 -- $e
done
cat $out
# the planner's block cost (VALU issue slots per generator block of eight reads) against the measured kernel time
for lib in miso_amd/libmiso_amd.so tools/_build/libmiso_swar.so; do
  for blk in 44 56 68 80 96; do
    run "hg19 block=$blk" --K 2 --reads-dist hg19 -- MISO_AMD_LIB=$lib MISO_K2_COST=$blk,1750,1170,840,720
    run "headline block=$blk" --K 2 -- MISO_AMD_LIB=$lib MISO_K2_COST=$blk,1750,1170,840,720
  done
done
cat $out
