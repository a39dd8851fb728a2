#!/bin/bash
# round 5: the collapsed step with G lanes per chain at 2 / 3 / 4 wavefronts per SIMD (register budgets 256 / 168 / 128):
# 40 000 chains at 4 lanes per chain are 2500 wavefronts -- one round only when a SIMD holds three of them
mkdir -p gpurun_out/r5
out=gpurun_out/r5/collapsed_ab.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
MISO_COLLAPSED_LANES=4 timeout 600 python -m pytest tests/test_gpu_collapsed.py -x -q > gpurun_out/r5/collapsed_tests.log 2>&1
echo "collapsed tests (4 lanes, 3 waves/SIMD) rc=$? $(tail -1 gpurun_out/r5/collapsed_tests.log)" >> $out
timeout 600 python -m pytest tests/test_gpu_convergent.py tests/test_gpu_collapsed.py -x -q > gpurun_out/r5/advice_tests.log 2>&1
echo "convergent + collapsed tests rc=$? $(tail -1 gpurun_out/r5/advice_tests.log)" >> $out
for lib in "" tools/_build/libmiso_k2c_w2.so tools/_build/libmiso_k2c_w4.so; do
  for g in 1 2 4 8; do
    [ -z "$lib" ] && run "in-tree(w3) G=$g" --collapsed 1 -- MISO_COLLAPSED_LANES=$g
    [ -n "$lib" ] && run "$lib G=$g" --collapsed 1 -- MISO_COLLAPSED_LANES=$g MISO_AMD_LIB=$lib
  done
done
run "in-tree defaults G=4" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_COLLAPSED_LANES=4
run "in-tree defaults G=1" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_COLLAPSED_LANES=1
run "in-tree hg19 G=4" --collapsed 1 --reads-dist hg19 -- MISO_COLLAPSED_LANES=4
run "in-tree hg19 G=1" --collapsed 1 --reads-dist hg19 -- MISO_COLLAPSED_LANES=1
cat $out
