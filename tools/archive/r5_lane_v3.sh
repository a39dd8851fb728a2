#!/bin/bash
# round 5: sampler_lane_ilp with the log-factorial table's head in LDS and the inversion search without divisions; the full
# GPU suite; then the two-isoform kernels with the iteration's sample stored BEHIND the Gibbs step (variant library)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lane_v3.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/lane_v3_tests.log 2>&1
echo "gpu tests rc=$? $(tail -1 gpurun_out/r5/lane_v3_tests.log)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/lane_v3_tests.log | head -10 >> $out
python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  run "ilp" --collapsed 1 -- MISO_X=0
  run "lean" --collapsed 1 -- MISO_LANE_ILP=0
done
for r in 100 300 3000; do run "ilp reads=$r" --collapsed 1 --reads $r -- MISO_X=0; run "lean reads=$r" --collapsed 1 --reads $r -- MISO_LANE_ILP=0; done
run "hg19 ilp" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "hg19 lean" --collapsed 1 --reads-dist hg19 -- MISO_LANE_ILP=0
run "defaults auto" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "hg19 defaults auto" --collapsed 1 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
V=tools/_build/libmiso_k2_storeafter.so
if [ -f $V ]; then
  MISO_AMD_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_heavy_tail.py -x -q > gpurun_out/r5/storeafter_tests.log 2>&1
  echo "store-after variant: parity / golden / heavy-tail tests rc=$? $(tail -1 gpurun_out/r5/storeafter_tests.log)" >> $out
  for rep in 1 2; do
    run "PE K=2 before" --K 2 --paired -- MISO_X=0
    run "PE K=2 after" --K 2 --paired -- MISO_AMD_LIB=$V
    run "headline before" --K 2 -- MISO_X=0
    run "headline after" --K 2 -- MISO_AMD_LIB=$V
  done
  run "PE K=2 hg19 before" --K 2 --paired --reads-dist hg19 -- MISO_X=0
  run "PE K=2 hg19 after" --K 2 --paired --reads-dist hg19 -- MISO_AMD_LIB=$V
  run "defaults before" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
  run "defaults after" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_AMD_LIB=$V
fi
cat $out
