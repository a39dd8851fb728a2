#!/bin/bash
# round 6: the paired-end size buckets' thresholds re-swept on the round's last kernels (hg19-like pair counts: K = 5 alone, whole-gene mix)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bc; mkdir -p $O
run() { echo "== $*" >> $O/ab.txt; env "$@" timeout 600 python tools/r6_clock.py pe_k5_hg19 pe_mix_hg19 --reps 3 --probe 0 2>&1 | grep -E "median" | cut -c1-60 >> $O/ab.txt; }
run MISO_X=0
for s in 1.5 3 4; do run MISO_PE_SHARE=$s; done
for t in 32 48 96 128; do run MISO_PE_T_WAVE=$t; done
for t in 128 192 384 512; do run MISO_PE_T_WIDE=$t; done
for t in 16 24 32 48; do run MISO_PE_T_32=$t; done
run MISO_X=0
cat $O/ab.txt
