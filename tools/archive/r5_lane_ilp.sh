#!/bin/bash
# round 5: sampler_lane with the Metropolis-Hastings step's transcendentals in five multi-argument stages, the binomial's
# set-up made beside them and its trials two per round (MISO_LANE_ILP=1, in-tree) against the one-call-after-the-other
# form (variant library); parity first.  Then the whole-gene mix with the 17-20 isoform class on 16 lanes.
mkdir -p gpurun_out/r5
out=gpurun_out/r5/lane_ilp.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_collapsed.py -x -q > gpurun_out/r5/lane_ilp_tests.log 2>&1
echo "collapsed tests rc=$? $(tail -1 gpurun_out/r5/lane_ilp_tests.log)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/lane_ilp_tests.log | head -10 >> $out
python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  run "collapsed ILP" --collapsed 1 -- MISO_X=0
  run "collapsed old" --collapsed 1 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
done
run "collapsed defaults ILP" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "collapsed defaults old" --collapsed 1 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
run "collapsed hg19 ILP" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "collapsed hg19 old" --collapsed 1 --reads-dist hg19 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
run "collapsed hg19 defaults ILP" --collapsed 1 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "collapsed hg19 defaults old" --collapsed 1 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_AMD_LIB=tools/_build/libmiso_lane_noilp.so
M="--K-range 3 20 --paired --events 16384"
run "mix (new lanes rule)" $M -- MISO_X=0
run "mix hg19 (new lanes rule)" $M --reads-dist hg19 -- MISO_X=0
run "mix hg19 T32=24" $M --reads-dist hg19 -- MISO_PE_T_32=24
run "mix T32=24" $M -- MISO_PE_T_32=24
cat $out
