#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest7.log 2>&1; tail -4 gpurun_out/r02/pytest7.log
MISO_FLAT_NC=7 timeout 900 python -m pytest tests -m gpu -x -q -k "parity or fuzz or golden or scale" > gpurun_out/r02/pytest7b.log 2>&1; tail -3 gpurun_out/r02/pytest7b.log
for K in 3 5 8 10 16 20; do
E=40000; [ $K -ge 16 ] && E=8192
python bench.py --no-cpu-baseline --no-matrix --steps 2 --K $K --events $E 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('K=$K', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
done
