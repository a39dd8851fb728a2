#!/bin/bash
# round 6: a workgroup-wide chain's Metropolis-Hastings step on four of its eight wavefronts (MISO_K2_WIDE_DEDUP=1 / 0): bit-exact
# tests on events of 20 ... 60 000 reads, the hg19-like rows, the wavefronts' times again
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q tests/test_gpu_heavy_tail.py tests/test_gpu_parity.py tests/test_gpu_scale.py 2>&1 | tail -3
for d in 1 0 1 0; do
  echo "== MISO_K2_WIDE_DEDUP=$d" >> $O/dedup.txt
  MISO_K2_WIDE_DEDUP=$d timeout 600 python tools/r6_clock.py se_k2_hg19 main --reps 5 --probe 0 2>&1 | grep -E "median" >> $O/dedup.txt
done
cat $O/dedup.txt
MISO_AMD_LIB=tools/_build/libmiso_wavetime.so timeout 600 python tools/archive/wave_time.py hg19 > $O/wave_time_hg19.txt 2>&1
tail -15 $O/wave_time_hg19.txt
