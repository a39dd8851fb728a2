#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest6.log 2>&1; tail -4 gpurun_out/r02/pytest6.log
for args in "" "--paired" "--chains 6 --iters 5000 --burn 500 --lag 10"; do
python bench.py --no-cpu-baseline --no-matrix --steps 4 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$args', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
done
