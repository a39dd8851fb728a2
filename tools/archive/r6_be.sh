#!/bin/bash
# round 6: which chains bound the launch of pe_k5_hg19 (diagnostic build -DMISO_GRP_WAVETIME)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6be; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT MISO_AMD_LIB=$GRAFT_REPO_ROOT/tools/_build/libmiso_grpwt.so
WT_K=5 WT_E=40000 timeout 600 python tools/archive/wave_time_grp.py timeline > $O/wave_time_k5_hg19.txt 2>&1
cat $O/wave_time_k5_hg19.txt | cut -c1-150
