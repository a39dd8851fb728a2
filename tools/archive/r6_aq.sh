#!/bin/bash
# round 6, late: re-profile the rows whose kernels changed (loop invariants, sampler_grp_all), then the closing run (tools/r6_last.sh)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6aq
ROUND=06 bash tools/round6_profiles.sh se_k5 se_k10 se_k5_hg19 pe_k5 pe_k10 pe_k5_hg19 pe_mix pe_mix_hg19 se_k2_defaults > gpurun_out/r6aq/profiles.log 2>&1
tail -3 gpurun_out/r6aq/profiles.log
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null
bash tools/r6_last.sh
