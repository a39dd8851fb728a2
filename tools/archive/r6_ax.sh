#!/bin/bash
# round 6, last: re-profile the two whole-gene rows (a segment per isoform count), then the closing run (tools/r6_last.sh)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6ax
ROUND=06 bash tools/round6_profiles.sh pe_mix pe_mix_hg19 > gpurun_out/r6ax/profiles.log 2>&1
tail -2 gpurun_out/r6ax/profiles.log
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null
bash tools/r6_last.sh
