#!/bin/bash
# round 4 closing run on the GPU box: the whole GPU suite, the generator's ceiling, the default bench line (what the driver
# runs) -> gpurun_out/r4/.  (The rocprofv3 passes: tools/round4_profiles.sh.)
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputests_final.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r4/gputests_final.log | tail -1)"
grep -E "^E|FAILED" gpurun_out/r4/gputests_final.log | head -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
hipcc --offload-arch=gfx950 -O3 tools/rng_bench.hip -o /tmp/rng_bench 2>/dev/null && /tmp/rng_bench > gpurun_out/r4/rng_bench.txt; cat gpurun_out/r4/rng_bench.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_default.out 2> gpurun_out/r4/bench_default.err
echo "bench rc=$?"; tail -c 3600 gpurun_out/r4/bench_default.out
cp gpurun_out/bench_full.json gpurun_out/r4/bench_default_full.json
