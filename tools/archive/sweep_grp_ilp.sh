#!/bin/bash
# Single-end class path: two Philox blocks in flight per lane up to TW = K - 1 <= MISO_GRP_ILP_MAX_TW.
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_grp_ilp.txt; : > $out
for tw in 0 3 7 15; do
  rm -f miso_amd/csrc/kernels_grp_c*.o
  make -s -j5 -C miso_amd/csrc EXTRA=-DMISO_GRP_ILP_MAX_TW=$tw 2>&1 | grep -v warning
  for k in 3 4 5 8 10; do
    r=$(python bench.py --no-cpu-baseline --K $k --steps 2 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"])')
    echo "ILP_MAX_TW=$tw K=$k $r" | tee -a $out
  done
done
rm -f miso_amd/csrc/kernels_grp_c*.o; make -s -j5 -C miso_amd/csrc 2>&1 | grep -v warning
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
