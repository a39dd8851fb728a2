#!/bin/bash
# round 5: eight lanes per chain in a whole-gene mix large enough that every class has two rounds of wavefronts
mkdir -p gpurun_out/r5
out=gpurun_out/r5/bigmix.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:60], d['roofline']['kernel_ms'])" >> $out
}
for ev in 16384 32768 65536; do
  for cfg in "--K-range 3 20 --paired --events $ev" "--K-range 3 20 --paired --events $ev --reads-dist hg19" "--K-range 3 8 --paired --events $ev"; do
    run "rule   $cfg" $cfg -- MISO_X=0
    run "rule in mixes $cfg" $cfg -- MISO_PE_LANES8=2
  done
done
cat $out
