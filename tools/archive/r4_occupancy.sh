#!/bin/bash
# round 4 experiments: (a) sampler_flat with one more workgroup per CU (register cap 128 / 168 instead of 168 / 256: spills against
# occupancy), (b) the two-isoform headline planned for three wavefronts per SIMD.  gpurun_out/r4/occupancy.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/occupancy.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
run "SE K=5" --K 5 -- MISO_X=0
run "SE K=5 4wg" --K 5 -- MISO_AMD_LIB=tools/_build/libmiso_flat4.so MISO_FLAT_WGS=4 MISO_FLAT_NC=8
run "SE K=5 4wg" --K 5 -- MISO_AMD_LIB=tools/_build/libmiso_flat4.so MISO_FLAT_WGS=4 MISO_FLAT_NC=7
run "SE K=3 4wg" --K 3 -- MISO_AMD_LIB=tools/_build/libmiso_flat4.so MISO_FLAT_WGS=4
run "SE K=3" --K 3 -- MISO_X=0
run "SE K=10" --K 10 -- MISO_X=0
run "SE K=10 3wg" --K 10 -- MISO_AMD_LIB=tools/_build/libmiso_flat12w3.so MISO_FLAT_WGS=3 MISO_FLAT_NC=4
run "SE K=10 3wg" --K 10 -- MISO_AMD_LIB=tools/_build/libmiso_flat12w3.so MISO_FLAT_WGS=3 MISO_FLAT_NC=5
run "SE K=10 3wg" --K 10 -- MISO_AMD_LIB=tools/_build/libmiso_flat12w3.so MISO_FLAT_WGS=3
run "headline" --K 2 -- MISO_X=0
run "headline 3/SIMD" --K 2 -- MISO_WAVE_SLOTS=3072 MISO_K2_WPB=4
run "headline 2.5/SIMD" --K 2 -- MISO_WAVE_SLOTS=2560 MISO_K2_WPB=4
run "headline wpb4" --K 2 -- MISO_K2_WPB=4
run "defaults" --K 2 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "defaults 3/SIMD" --K 2 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_WAVE_SLOTS=3072
cat $out
