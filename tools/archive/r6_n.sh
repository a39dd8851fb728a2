#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
MISO_AMD_LIB=tools/_build/libmiso_wavetime.so timeout 600 python tools/archive/wave_time.py hg19 > $O/wave_time_hg19.txt 2>&1
tail -20 $O/wave_time_hg19.txt
MISO_AMD_LIB=tools/_build/libmiso_wavetime.so MISO_K2_BALANCE=0 timeout 600 python tools/archive/wave_time.py hg19 > $O/wave_time_hg19_nobalance.txt 2>&1
tail -16 $O/wave_time_hg19_nobalance.txt
