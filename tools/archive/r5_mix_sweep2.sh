#!/bin/bash
# round 5: lanes per chain of the 17-20 isoform class and the size buckets' share factor, uniform and hg19-like pair counts
mkdir -p gpurun_out/r5
out=gpurun_out/r5/mix_sweep2.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:100], d['roofline']['kernel_ms'])" >> $out
}
M="--K-range 3 20 --paired --events 16384"
L16=MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:16
for sh in 2.0 1.5 1.0 0.75; do
  run "mix share=$sh" $M -- MISO_PE_SHARE=$sh
  run "mix share=$sh 16 lanes" $M -- MISO_PE_SHARE=$sh $L16
  run "mix hg19 share=$sh" $M --reads-dist hg19 -- MISO_PE_SHARE=$sh
  run "mix hg19 share=$sh 16 lanes" $M --reads-dist hg19 -- MISO_PE_SHARE=$sh $L16
done
run "K=17..20 20000 default" --K-range 17 20 --paired --events 20000 -- MISO_X=0
run "K=17..20 20000 16 lanes" --K-range 17 20 --paired --events 20000 -- MISO_GENERAL_LANES=16 MISO_NO_PE_BUCKETS=1
run "K=17..20 20000 32 lanes" --K-range 17 20 --paired --events 20000 -- MISO_GENERAL_LANES=32 MISO_NO_PE_BUCKETS=1
cat $out
