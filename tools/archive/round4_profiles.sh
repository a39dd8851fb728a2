#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes (kernel trace, SQ / FETCH / WRITE counters) of the bench line's
# workloads -- the headline and every matrix row -- -> gpurun_out/prof_<tag>/, summaries r${ROUND:-04}_<tag>_summary.txt, and
# the tables bench.py reads (profiles/traffic.json, profiles/valu_model.json; copy them back into profiles/).
#   tools/round4_profiles.sh [tag ...]      (default: all)
set -u
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out
mkdir -p $OUT
ALL="se_k2: se_k2_defaults:--chains_6_--iters_5000_--burn_500_--lag_10 se_k5:--K_5 se_k10:--K_10 pe_k2:--paired pe_k5:--paired_--K_5 pe_k10:--paired_--K_10_--events_20000 pe_mix:--paired_--K-range_3_20_--events_16384 se_k2_hg19:--reads-dist_hg19 se_k2_hg19_defaults:--reads-dist_hg19_--chains_6_--iters_5000_--burn_500_--lag_10 pe_k2_hg19:--reads-dist_hg19_--paired se_k5_hg19:--reads-dist_hg19_--K_5 pe_k5_hg19:--reads-dist_hg19_--paired_--K_5 pe_mix_hg19:--reads-dist_hg19_--paired_--K-range_3_20_--events_16384 se_k2_collapsed:--collapsed_1 se_k2_defaults_collapsed:--collapsed_1_--chains_6_--iters_5000_--burn_500_--lag_10 se_k2_hg19_collapsed:--collapsed_1_--reads-dist_hg19 se_k2_hg19_defaults_collapsed:--collapsed_1_--reads-dist_hg19_--chains_6_--iters_5000_--burn_500_--lag_10"
for tag_args in $ALL; do
  tag=${tag_args%%:*}; args=${tag_args#*:}; args=${args//_/ }
  if [ $# -gt 0 ] && [[ ! " $* " =~ " $tag " ]]; then continue; fi
  bash tools/profile.sh $tag --no-streams $args > /dev/null 2>&1
  python3 tools/prof_summary.py $OUT/prof_$tag $OUT/r${ROUND:-04}_${tag}_summary.txt > $OUT/r${ROUND:-04}_${tag}_summary.log 2>&1
  tail -2 $OUT/r${ROUND:-04}_${tag}_summary.log
  rm -rf $OUT/prof_$tag/*/*.db $OUT/prof_$tag/trace $OUT/prof_$tag/pmc_*   # keep the text, drop the databases
done
cp profiles/traffic.json profiles/valu_model.json $OUT/ 2>/dev/null
