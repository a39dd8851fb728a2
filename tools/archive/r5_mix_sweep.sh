#!/bin/bash
# round 5: where the whole-gene paired-end mix loses against the sum of its classes: lanes per class, buckets, kernels one
# after the other, hardware queues; then the per-chain wave-time timeline (variant build -DMISO_GRP_WAVETIME)
mkdir -p gpurun_out/r5
out=gpurun_out/r5/mix_sweep.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:100], d['roofline']['kernel_ms'])" >> $out
}
M="--K-range 3 20 --paired --events 16384"
run "mix default" $M -- MISO_X=0
run "mix serial kernels" $M -- MISO_SERIAL_KERNELS=1
run "mix no buckets" $M -- MISO_NO_PE_BUCKETS=1
run "mix no buckets, 16 lanes everywhere" $M -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:16
run "mix no buckets, 16 up to 16, 32 beyond" $M -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:32
run "mix no buckets, 32 everywhere" $M -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES_BY_CLASS=4:32,8:32,12:32,16:32,32:32
run "mix no buckets, 16 lanes, serial" $M -- MISO_NO_PE_BUCKETS=1 MISO_SERIAL_KERNELS=1 MISO_GENERAL_LANES_BY_CLASS=4:16,8:16,12:16,16:16,32:16
run "mix 2 hw queues" $M -- GPU_MAX_HW_QUEUES=2
run "mix 4 hw queues" $M -- GPU_MAX_HW_QUEUES=4
run "mix multi" $M -- MISO_PE_MULTI=1
# single classes at the mix's sizes: 16384 genes x 4/18 (2/18 for 3-4 isoforms)
run "K=3..4 alone" --K-range 3 4 --paired --events 1820 -- MISO_X=0
run "K=5..8 alone" --K-range 5 8 --paired --events 3641 -- MISO_X=0
run "K=9..12 alone" --K-range 9 12 --paired --events 3641 -- MISO_X=0
run "K=13..16 alone" --K-range 13 16 --paired --events 3641 -- MISO_X=0
run "K=17..20 alone" --K-range 17 20 --paired --events 3641 -- MISO_X=0
run "K=17..20 alone 16 lanes" --K-range 17 20 --paired --events 3641 -- MISO_NO_PE_BUCKETS=1 MISO_GENERAL_LANES=16
cat $out
if [ -f tools/_build/libmiso_grpwt.so ]; then
  MISO_AMD_LIB=tools/_build/libmiso_grpwt.so timeout 600 python tools/wave_time_grp.py uniform timeline > gpurun_out/r5/mix_timeline.txt 2>&1
  head -60 gpurun_out/r5/mix_timeline.txt
fi
