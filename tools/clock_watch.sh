#!/bin/bash
# Sample the shader clock and socket power while the headline bench runs (is the kernel power-limited?)
cd "$(dirname "$0")/.."
python bench.py --no-cpu-baseline --steps 40 > gpurun_out/clock_bench.json 2>&1 &
pid=$!
: > gpurun_out/clock_watch.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' ' >> gpurun_out/clock_watch.txt
  echo >> gpurun_out/clock_watch.txt
  sleep 0.2
done
wait $pid
tail -c 600 gpurun_out/clock_bench.json
sort gpurun_out/clock_watch.txt | uniq -c | sort -rn | head -12
