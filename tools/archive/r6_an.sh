#!/bin/bash
# round 6: the build with the record registers named in pe_dense's wait and packed sampler_flat launches sized for three workgroups per CU
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6an; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
timeout 900 python tools/archive/r6_shape.py K=7,paired=1 K=8,paired=1 K=5,reads=hg19 K=6,reads=hg19 K=8,reads=hg19 K=6 --reps 2 2>&1 | grep median > $O/shapes.txt
MISO_FLAT_PACK_WGS2=1 timeout 900 python tools/archive/r6_shape.py K=5,reads=hg19 K=6,reads=hg19 --reps 2 2>&1 | grep median >> $O/shapes.txt
cat $O/shapes.txt
timeout 900 python tools/r6_clock.py pe_mix pe_mix_hg19 pe_k5_hg19 --reps 3 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 > $O/rows.txt
cat $O/rows.txt
