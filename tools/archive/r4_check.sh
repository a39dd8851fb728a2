#!/bin/bash
# round 4: full GPU test suite + the rows the round's kernel changes touch; gpurun_out/r4/check.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/check.txt; : > $out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r4/gputests.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r4/gputests.log | head -20 >> $out
python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for K in 3 4 5 7 8 10 12 16 20; do run "SE K=$K" --K $K -- MISO_X=0; done
run "SE K=5 hg19" --K 5 --reads-dist hg19 -- MISO_X=0
run "SE mix 3-20" --K-range 3 20 -- MISO_X=0
run "SE K=40" --K 40 --events 8192 -- MISO_X=0
run "PE K=40" --K 40 --paired --events 4096 -- MISO_X=0
run "headline" --K 2 -- MISO_X=0
cat $out
