#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
ROUND=06 bash tools/round6_profiles.sh se_k2 se_k2_hg19 > gpurun_out/r6p/profiles.log 2>&1
tail -2 gpurun_out/r6p/profiles.log
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null
bash tools/r6_last.sh
