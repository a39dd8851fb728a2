#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench matrix -> gpurun_out/r02_configs.jsonl
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
: > $OUT/r02_configs.jsonl
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matrix "$@" | tail -1 >> $OUT/r02_configs.jsonl; }
run                                   # configs[1]: SE K=2, 1 chain, 2500+5000
run --chains 6 --iters 5000 --burn 500 --lag 10      # MISO defaults
for K in 3 4 5 6 8 10 12; do run --K $K; done
run --K 16 --events 8192; run --K 20 --events 8192
run --paired                           # configs[2]: PE K=2
for K in 3 4 5 8; do run --paired --K $K; done
for K in 10 16 20; do run --paired --K $K --events 20000; done
run --K-range 3 20                                     # configs[3] proxy: mixed batch, concurrent kernels
run --paired --K-range 3 20 --events 16384
run --paired --K-range 3 20 --events 8192
python3 - <<'PY'
import json
for l in open("gpurun_out/r02_configs.jsonl"):
    d = json.loads(l); c = d["config"]
    print("%-40s K=%-8s chains=%d events=%-6d %10.1f events/s  %8.1f ms" % (d["roofline"]["kernel"][:40], c["K"], c["chains"], c["events_per_gpu"], d["value"], d["roofline"]["kernel_ms"]))
PY
