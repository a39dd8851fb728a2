#!/bin/bash
for K in 3 4 5 6 8; do
for wgs in 3 2; do
MISO_NO_AUTOTUNE=1 MISO_FLAT_WGS=$wgs python bench.py --no-cpu-baseline --no-matrix --steps 2 --K $K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('K=$K wgs=$wgs', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
done; done
