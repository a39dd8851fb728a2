#!/bin/bash
# round 4 closing run on the GPU box: the whole GPU suite, the default bench line (what the driver runs), then the rocprofv3
# passes of the rows whose kernels changed this round (+ the headline) -> gpurun_out/r4/, profiles to copy back
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests_final.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r4/gputests_final.log | tail -1)"
grep -E "^E|FAILED" gpurun_out/r4/gputests_final.log | head -10
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_default.out 2> gpurun_out/r4/bench_default.err
echo "bench rc=$?"; tail -c 3600 gpurun_out/r4/bench_default.out
cp gpurun_out/bench_full.json gpurun_out/r4/bench_default_full.json
ROUND=04 timeout 1500 bash tools/round4_profiles.sh se_k2 se_k5 se_k10 se_k5_hg19 pe_k10 2>&1 | tail -12
