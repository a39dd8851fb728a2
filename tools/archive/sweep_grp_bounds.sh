#!/bin/bash
# Workgroups per CU promised to the compiler for the small-K general kernels (register budget) x K.
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_grp_bounds.txt; : > $out
for mb in 2 3 4; do
  rm -f miso_amd/csrc/kernels_grp_c4.o miso_amd/csrc/kernels_grp_c8.o
  make -s -j4 -C miso_amd/csrc EXTRA=-DMISO_GRP_MINBLOCKS=$mb 2>&1 | grep -v warning
  for k in 3 4 5 8; do
    r=$(python bench.py --no-cpu-baseline --K $k --steps 2 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"])')
    echo "MINBLOCKS=$mb K=$k $r" | tee -a $out
  done
done
rm -f miso_amd/csrc/kernels_grp_c4.o miso_amd/csrc/kernels_grp_c8.o; make -s -j4 -C miso_amd/csrc 2>&1 | grep -v warning
