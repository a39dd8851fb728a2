#!/bin/bash
# round 6: which genes bound the whole-gene paired-end launch (diagnostic build -DMISO_GRP_WAVETIME, tools/archive/wave_time_grp.py)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT MISO_AMD_LIB=tools/_build/libmiso_grpwt.so
timeout 600 python tools/archive/wave_time_grp.py timeline > $O/wave_time_mix_hg19.txt 2>&1
cat $O/wave_time_mix_hg19.txt | cut -c1-150
timeout 600 python tools/archive/wave_time_grp.py uniform timeline > $O/wave_time_mix_uniform.txt 2>&1
head -32 $O/wave_time_mix_uniform.txt | cut -c1-150
