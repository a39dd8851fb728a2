#!/bin/bash
# A/B of the paired-end read loops for more than two isoforms: dense records (pe_dense) vs the plain
# records' quad loops (MISO_NO_PE_DENSE=1).  Run on the GPU box; prints events/s per isoform count.
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"; }
for K in ${KS:-3 4 5 8 10 16 20}; do
  E=${EVENTS:-$((K <= 8 ? 40000 : 20000))}
  python bench.py --no-cpu-baseline --no-matrix --paired --K $K --events $E --steps 2 2>/dev/null | line "PE K=$K dense"
  [ -n "$NO_AB" ] || MISO_NO_PE_DENSE=1 python bench.py --no-cpu-baseline --no-matrix --paired --K $K --events $E --steps 2 2>/dev/null | line "PE K=$K plain"
done
python bench.py --no-cpu-baseline --no-matrix --paired --K-range 3 20 --events ${MIXED_EVENTS:-16384} --steps 2 2>/dev/null | line "PE K=3..20 dense"
[ -n "$NO_AB" ] || MISO_NO_PE_DENSE=1 python bench.py --no-cpu-baseline --no-matrix --paired --K-range 3 20 --events ${MIXED_EVENTS:-16384} --steps 2 2>/dev/null | line "PE K=3..20 plain"
