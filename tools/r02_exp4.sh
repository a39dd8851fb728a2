#!/bin/bash
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest4.log 2>&1; tail -3 gpurun_out/r02/pytest4.log
for mode in delta_auto delta_g8 delta_g4 plain plain_g8; do
  unset MISO_NO_PE_DELTA MISO_LANES_PER_CHAIN
  case $mode in plain*) export MISO_NO_PE_DELTA=1;; esac
  case $mode in *_g8) export MISO_LANES_PER_CHAIN=8;; *_g4) export MISO_LANES_PER_CHAIN=4;; esac
  python bench.py --no-cpu-baseline --no-matrix --paired --steps 2 2>gpurun_out/r02/pe_$mode.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('PE K=2 $mode', d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" || tail -5 gpurun_out/r02/pe_$mode.err
done
