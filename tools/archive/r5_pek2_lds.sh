#!/bin/bash
# round 5: paired-end two-isoform events: the LDS budget per workgroup (chains per wavefront against workgroups per CU)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/pek2_lds.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 2 --paired" "--K 2 --paired --reads-dist hg19"; do
  for kb in 40 52 64 80 100 150; do run "$cfg" $cfg -- MISO_LDS_MAX_KB=$kb; done
done
cat $out
