#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6ad
ROUND=06 bash tools/round6_profiles.sh se_k5_hg19 > gpurun_out/r6ad/profiles.log 2>&1
tail -2 gpurun_out/r6ad/profiles.log | cut -c1-250
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null
bash tools/r6_last.sh
