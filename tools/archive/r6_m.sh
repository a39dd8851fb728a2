#!/bin/bash
# round 6: how long every wavefront of the hg19-like two-isoform launch ran (diagnostic build), and the planner under other cost models
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
MISO_AMD_LIB=tools/_build/libmiso_wavetime.so timeout 600 python tools/archive/wave_time.py hg19 > $O/wave_time_hg19.txt 2>&1
cat $O/wave_time_hg19.txt | tail -20
MISO_AMD_LIB=tools/_build/libmiso_wavetime.so MISO_K2_BALANCE=0 timeout 600 python tools/archive/wave_time.py hg19 > $O/wave_time_hg19_nobalance.txt 2>&1
cat $O/wave_time_hg19_nobalance.txt | tail -16
for c in "" "54,1750,1170,730,610" "54,1750,1100,700,600" "50,1700,1100,700,600"; do
  echo "== MISO_K2_COST=$c" >> $O/cost.txt
  if [ -z "$c" ]; then timeout 600 python tools/r6_clock.py main se_k2_hg19 se_k2_hg19_defaults --reps 4 --probe 0 2>&1 | grep -E "median" >> $O/cost.txt
  else MISO_K2_COST=$c timeout 600 python tools/r6_clock.py main se_k2_hg19 se_k2_hg19_defaults --reps 4 --probe 0 2>&1 | grep -E "median" >> $O/cost.txt; fi
done
cat $O/cost.txt
