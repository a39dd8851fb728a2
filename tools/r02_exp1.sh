#!/bin/bash
# round-2 experiment 1: issue costs, parity of the paired launch, paired vs unpaired headline, placement
mkdir -p gpurun_out/r02
./tools/bin/issue_bench > gpurun_out/r02/issue_costs.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest1.log 2>&1; tail -3 gpurun_out/r02/pytest1.log
python bench.py --no-cpu-baseline --no-matrix --steps 5 > gpurun_out/r02/bench_pair.json 2> gpurun_out/r02/bench_pair.err
MISO_K2_PAIR=0 python bench.py --no-cpu-baseline --no-matrix --steps 5 > gpurun_out/r02/bench_nopair.json 2> gpurun_out/r02/bench_nopair.err
python tools/placement_check.py > gpurun_out/r02/placement_pair.txt 2>&1
MISO_K2_PAIR=0 python tools/placement_check.py > gpurun_out/r02/placement_nopair.txt 2>&1
for f in bench_pair bench_nopair; do python -c "
import json,sys
d=json.load(open('gpurun_out/r02/$f.json')); print('$f', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline'].get('rng_frac'))" || tail -5 gpurun_out/r02/$f.err; done
cat gpurun_out/r02/placement_pair.txt gpurun_out/r02/placement_nopair.txt
cat gpurun_out/r02/issue_costs.txt
