#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 passes of ONE bench configuration and their summary.
#   tools/profile_one.sh <tag> [bench args...]   -> gpurun_out/r02_<tag>_summary.txt, traffic.json, valu_model.json
set -u
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out; mkdir -p $OUT
tag=$1; shift
bash tools/profile.sh $tag "$@" > /dev/null 2>&1
python3 tools/prof_summary.py $OUT/prof_$tag $OUT/r02_${tag}_summary.txt > $OUT/r02_${tag}_summary.log 2>&1
tail -2 $OUT/r02_${tag}_summary.log
rm -rf $OUT/prof_$tag/*/*.db $OUT/prof_$tag/trace $OUT/prof_$tag/pmc_*
cp profiles/traffic.json profiles/valu_model.json $OUT/ 2>/dev/null
