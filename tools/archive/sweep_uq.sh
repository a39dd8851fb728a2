#!/bin/bash
# Sweep the Philox blocks in flight per lane (MISO_K2_UQ) x lanes per chain for the headline bench.
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_uq.txt; : > $out
for uq in 2 3 4 6 8; do
  rm -f miso_amd/csrc/kernels_k2.o
  make -s -C miso_amd/csrc EXTRA=-DMISO_K2_UQ=$uq || exit 1
  for g in 2 3 4; do
    r=$(MISO_LANES_PER_CHAIN=$g python bench.py --no-cpu-baseline --steps 2 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["roofline"]["kernel_ms"])')
    echo "UQ=$uq G=$g $r" | tee -a $out
  done
done
rm -f miso_amd/csrc/kernels_k2.o; make -s -C miso_amd/csrc
