#!/bin/bash
# round 4: sampler_flatl (Metropolis-Hastings by one wavefront per workgroup, one chain per lane) -- parity against the oracle
# with the new kernel forced on, then timing against sampler_flat; gpurun_out/r4/flatl.txt
mkdir -p gpurun_out/r4
out=gpurun_out/r4/flatl.txt; : > $out
export MISO_FLAT_LANE_MH=1
timeout 900 python -m pytest -x -q tests/test_gpu_heavy_tail.py -k "three_or_more_isoforms_bit_exact" tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py > gpurun_out/r4/flatl_parity.log 2>&1
echo "parity rc=$? $(tail -1 gpurun_out/r4/flatl_parity.log)" >> $out
if ! grep -q "passed" gpurun_out/r4/flatl_parity.log || grep -q "failed" gpurun_out/r4/flatl_parity.log; then tail -40 gpurun_out/r4/flatl_parity.log; cat $out; exit 1; fi
run() {  # K extra-env...
  K=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams --K $K --events 40000 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('K=$K $*', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $out
}
for K in 5 10; do
  run $K MISO_FLAT_LANE_MH=0
  run $K MISO_FLAT_LANE_MH=1
  for nc in 6 8 10 12 16; do run $K MISO_FLAT_LANE_MH=1 MISO_FLAT_NC=$nc; done
done
run 5 MISO_FLAT_LANE_MH=1 MISO_FLAT_NC=8 MISO_FLAT_WGS=4
run 5 MISO_FLAT_LANE_MH=1 MISO_FLAT_NC=10 MISO_FLAT_WGS=2
run 10 MISO_FLAT_LANE_MH=1 MISO_FLAT_NC=6 MISO_FLAT_WGS=3
for K in 3 4 7 16; do run $K MISO_FLAT_LANE_MH=0; run $K MISO_FLAT_LANE_MH=1; done
cat $out
