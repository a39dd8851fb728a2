#!/bin/bash
# round 4: the planner's cost model (plan.hpp k2_cost_single) against the lazy-low-bits read loop: forced single-width launches
# (tools/sweep_multi.py calib) and the hg19-like rows under scaled step costs; gpurun_out/r4/cost_sweep.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/cost_sweep.txt; : > $out
python tools/sweep_multi.py calib 2>/dev/null >> $out
run() {
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for cost in 54,1750,1170,840,720 54,1400,940,670,580 54,2200,1460,1050,900 54,1750,1400,1100,1000 54,1750,1000,650,520 40,1750,1170,840,720; do
  run "hg19" --K 2 --reads-dist hg19 -- MISO_K2_COST=$cost
  run "hg19 defaults" --K 2 --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_K2_COST=$cost
  run "headline" --K 2 -- MISO_K2_COST=$cost
done
cat $out
