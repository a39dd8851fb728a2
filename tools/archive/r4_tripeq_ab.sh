#!/bin/bash
# round 4: the two-isoform read loop's test for a high half on the threshold once per trip (MISO_K2_TRIP_EQ=1 variant)
# against once per block (product); gpurun_out/r4/tripeq_ab.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/tripeq_ab.txt; : > $out
MISO_AMD_LIB=tools/_build/libmiso_tripeq.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_heavy_tail.py tests/test_gpu_collapsed.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4/gputests_tripeq.log 2>&1
echo "gpu tests (variant) rc=$? $(grep -E 'passed|failed' gpurun_out/r4/gputests_tripeq.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r4/gputests_tripeq.log | head -20 >> $out
run() {
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  for e in MISO_X=0 MISO_AMD_LIB=tools/_build/libmiso_tripeq.so; do
    run "headline" --K 2 -- $e
    run "defaults" --K 2 --chains 6 --iters 5000 --burn 500 --lag 10 -- $e
    run "hg19" --K 2 --reads-dist hg19 -- $e
    run "hg19 defaults" --K 2 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- $e
  done
done
cat $out
