#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench matrix (-> gpurun_out/configs.jsonl) and the
# rocprofv3 profiles of the main configurations (-> gpurun_out/prof_<tag>/ + summaries).
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
: > $OUT/configs.jsonl
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" | tail -1 >> $OUT/configs.jsonl; }
run                                   # configs[1]: SE K=2, 1 chain, 2500+5000
run --chains 6 --iters 5000 --burn 500 --lag 10      # MISO defaults
for K in 3 4 5 8 10; do run --K $K; done
run --K 16 --events 8192
run --paired                           # configs[2]: PE K=2
for K in 3 5 8 10; do run --paired --K $K --events 8192; done
run --K 18 --events 8192; run --K 20 --events 8192      # K <= 32 kernel, 35-40 read classes
run --K-range 3 20                                     # configs[3] proxy: mixed batch, concurrent kernels
run --paired --K-range 3 20 --events 8192
for tag_args in "se_k2:" "se_k3:--K 3" "pe_k2:--paired" "pe_k3:--paired --K 3 --events 8192"; do
  tag=${tag_args%%:*}; args=${tag_args#*:}
  bash tools/profile.sh $tag $args > /dev/null 2>&1
  python3 tools/prof_summary.py $OUT/prof_$tag $OUT/r01_${tag}_summary.txt > /dev/null 2>&1
done
cp profiles/traffic.json $OUT/traffic.json 2>/dev/null
python3 - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l); c = d["config"]
    print("%-28s K=%-7s chains=%d events=%-6d %10.1f events/s  %8.1f ms" % (d["roofline"]["kernel"][:28], c["K"], c["chains"], c["events_per_gpu"], d["value"], d["roofline"]["kernel_ms"]))
PY
