#!/bin/bash
# round 6: sampler_flat's compile-time knobs again on the faster kernel (chunk width of the leader sums, Philox blocks per read-loop trip, four workgroups per CU)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6al; mkdir -p $O
for v in "" ch8 uq4 uq1 wgs4 wgs4:MISO_FLAT_WGS=4 wgs4:MISO_FLAT_WGS=3; do
  lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="MISO_X=0"
  so=miso_amd/libmiso_amd.so; [ -n "$lib" ] && so=tools/_build/libmiso_$lib.so
  echo "== $v" >> $O/ab.txt
  env $envs MISO_AMD_LIB=$GRAFT_REPO_ROOT/$so timeout 600 python tools/r6_clock.py se_k5 se_k10 se_k5_hg19 --reps 2 --probe 0 2>&1 | grep -E "median" | cut -c1-60 >> $O/ab.txt
done
cat $O/ab.txt
