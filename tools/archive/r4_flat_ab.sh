#!/bin/bash
# round 4: sampler_flat A/B -- thresholds only for the chains whose psi changed (default) against every chain every step
# (libmiso_thrall.so), four against three workgroups per CU up to four isoforms (libmiso_kc4w3.so).  gpurun_out/r4/flat_ab.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/flat_ab.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for K in 3 4 5 6 7 8 10 12; do
  run "SE K=$K skip" --K $K -- MISO_X=0
  run "SE K=$K all " --K $K -- MISO_AMD_LIB=tools/_build/libmiso_thrall.so
done
for K in 3 4; do
  run "SE K=$K 3wg skip" --K $K -- MISO_AMD_LIB=tools/_build/libmiso_kc4w3.so MISO_FLAT_WGS=3
done
run "SE K=5 hg19 skip" --K 5 --reads-dist hg19 -- MISO_X=0
run "SE K=5 hg19 all " --K 5 --reads-dist hg19 -- MISO_AMD_LIB=tools/_build/libmiso_thrall.so
cat $out
