#!/bin/bash
# round 4: the scalar step by one wavefront per workgroup (lane_mh.hpp): sampler_flatl (single-end) and sampler_pel (paired-end,
# records in LDS).  Parity against the oracle with the new kernels forced on, then timing; gpurun_out/r4/lane_mh.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/lane_mh.txt; : > $out
par() {  # name env... -- pytest args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python -m pytest -x -q "$@" > gpurun_out/r4/parity_$name.log 2>&1
  echo "parity $name rc=$? $(tail -1 gpurun_out/r4/parity_$name.log)" >> $out
}
par flatl_a MISO_FLAT_LANE_MH=1 -- tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py
par flatl_b MISO_FLAT_LANE_MH=1 -- tests/test_gpu_heavy_tail.py -k three_or_more_isoforms_bit_exact
par pel_a MISO_PEL=1 -- tests/test_gpu_parity.py tests/test_gpu_paired_dense.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py
par pel_b MISO_PEL=1 -- tests/test_gpu_heavy_tail.py -k pair_counts_three_or_more
par pel_c MISO_PEL=1 MISO_PEL_STAB=1 -- tests/test_gpu_paired_dense.py tests/test_gpu_parity.py -k paired
cat $out
if grep -q "failed\|error" $out; then for f in gpurun_out/r4/parity_*.log; do echo "== $f"; tail -30 $f; done; fi
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for K in 5 10; do
  run "SE K=$K" --K $K -- MISO_FLAT_LANE_MH=0
  for nc in 8 10; do run "SE K=$K" --K $K -- MISO_FLAT_LANE_MH=1 MISO_FLAT_NC=$nc; done
done
for K in 5 10 3; do
  E=40000; [ $K -ge 9 ] && E=20000
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=0
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=1
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=1 MISO_PEL_NO_REC=1
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=1 MISO_PEL_STAB=1
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=1 MISO_GENERAL_LANES=32
  run "PE K=$K" --K $K --paired --events $E -- MISO_PEL=0 MISO_GENERAL_LANES=32
done
run "PE mix" --K-range 3 20 --paired --events 16384 -- MISO_PEL=0
run "PE mix" --K-range 3 20 --paired --events 16384 -- MISO_PEL=1
if [ -f tools/_build/libmiso_prof.so ]; then
  for lmh in 0 1; do
    MISO_AMD_LIB=tools/_build/libmiso_prof.so MISO_FLAT_LANE_MH=$lmh EVENTS=40000 KS=5,10 NCS=8 timeout 300 python tools/phase_prof_flat.py >> $out 2>&1
  done
fi
cat $out
