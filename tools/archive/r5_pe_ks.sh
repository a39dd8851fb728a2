#!/bin/bash
# round 5: paired-end K = 5 with a kernel that carries only that count's read loop (MISO_PE_ONLY_K) and, beside it, the slice
# layout at compile time (MISO_GRP_KS_FIXED) -- variant libraries, valid for K = 5 only -- against the class kernel
mkdir -p gpurun_out/r5
out=gpurun_out/r5/pe_ks.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2; do
  run "class kernel   K=5" --K 5 --paired -- MISO_X=0
  run "only K=5       K=5" --K 5 --paired -- MISO_AMD_LIB=tools/_build/libmiso_peonly5.so
  run "only K=5 + ks  K=5" --K 5 --paired -- MISO_AMD_LIB=tools/_build/libmiso_peks5.so
done
run "class kernel   K=5 hg19" --K 5 --paired --reads-dist hg19 -- MISO_X=0
run "only K=5 + ks  K=5 hg19" --K 5 --paired --reads-dist hg19 -- MISO_AMD_LIB=tools/_build/libmiso_peks5.so
cat $out
