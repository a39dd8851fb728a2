#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6zz; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT MISO_AMD_LIB=tools/_build/libmiso_flatwt.so
timeout 600 python tools/archive/wave_time_flat.py > $O/wave_time_flat_hg19.txt 2>&1; cat $O/wave_time_flat_hg19.txt
timeout 600 python tools/archive/wave_time_flat.py uniform > $O/wave_time_flat_uniform.txt 2>&1; cat $O/wave_time_flat_uniform.txt
