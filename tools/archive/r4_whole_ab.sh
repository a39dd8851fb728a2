#!/bin/bash
# round 4: the descriptor read loop's whole-unit fast path (default) against the masked path for every unit (libmiso_nowhole.so),
# same box, alternating; gpurun_out/r4/whole_ab.txt.  Parity first (the unit order changed).
mkdir -p gpurun_out/r4
out=gpurun_out/r4/whole_ab.txt; : > $out
timeout 600 python -m pytest -x -q tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py tests/test_gpu_heavy_tail.py > gpurun_out/r4/whole_parity.log 2>&1
echo "parity rc=$? $(grep -E 'passed|failed' gpurun_out/r4/whole_parity.log | tail -1)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
for K in 4 5 7 10 16; do
  run "SE K=$K whole  " --K $K -- MISO_X=0
  run "SE K=$K nowhole" --K $K -- MISO_AMD_LIB=tools/_build/libmiso_nowhole.so
done
done
run "SE K=5 hg19 whole  " --K 5 --reads-dist hg19 -- MISO_X=0
run "SE K=5 hg19 nowhole" --K 5 --reads-dist hg19 -- MISO_AMD_LIB=tools/_build/libmiso_nowhole.so
run "SE mix whole  " --K-range 3 20 -- MISO_X=0
run "SE mix nowhole" --K-range 3 20 -- MISO_AMD_LIB=tools/_build/libmiso_nowhole.so
cat $out
