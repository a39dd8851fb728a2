#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes (kernel trace, SQ / FETCH / WRITE counters) of the
# bench line's workloads -> gpurun_out/prof_<tag>/, summaries r02_<tag>_summary.txt, and the tables
# bench.py reads (profiles/traffic.json, profiles/valu_model.json; copy them back into profiles/).
set -u
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out
mkdir -p $OUT
for tag_args in "se_k2:" "se_k2_defaults:--chains 6 --iters 5000 --burn 500 --lag 10" "se_k5:--K 5" "se_k10:--K 10" "pe_k2:--paired" "pe_k5:--paired --K 5" "pe_k10:--paired --K 10 --events 20000"; do
  tag=${tag_args%%:*}; args=${tag_args#*:}
  bash tools/profile.sh $tag $args > /dev/null 2>&1
  python3 tools/prof_summary.py $OUT/prof_$tag $OUT/r02_${tag}_summary.txt > $OUT/r02_${tag}_summary.log 2>&1
  tail -2 $OUT/r02_${tag}_summary.log
  rm -rf $OUT/prof_$tag/*/*.db $OUT/prof_$tag/trace $OUT/prof_$tag/pmc_*   # keep the text, drop the databases
done
cp profiles/traffic.json profiles/valu_model.json $OUT/ 2>/dev/null
