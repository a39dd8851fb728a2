#!/bin/bash
# round-2 experiment 2: sampler_flat parity (NC = 1 and forced NC) and first timings
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest2a.log 2>&1; tail -5 gpurun_out/r02/pytest2a.log
MISO_FLAT_NC=7 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest2b.log 2>&1; tail -5 gpurun_out/r02/pytest2b.log
for K in 3 5 10; do
  python bench.py --no-cpu-baseline --no-matrix --K $K --steps 2 > gpurun_out/r02/flat_k$K.json 2> gpurun_out/r02/flat_k$K.err
  python -c "
import json
d=json.load(open('gpurun_out/r02/flat_k$K.json')); print('K=$K', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" || tail -5 gpurun_out/r02/flat_k$K.err
done
