#!/bin/bash
# round 6: sampler_flat<KC, KS, UNI> (every event of the launch has KS isoforms) against MISO_FLAT_NO_UNI=1; sampler_grp without machine-level hoisting
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ah; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "flat or parity or fuzz or heavy or wide or convergent" > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for v in 0 1 0 1; do
  echo "== MISO_FLAT_NO_UNI=$v" >> $O/ab.txt
  if [ $v = 1 ]; then export MISO_FLAT_NO_UNI=1; else unset MISO_FLAT_NO_UNI; fi
  timeout 900 python tools/r6_clock.py se_k5 se_k10 se_k5_hg19 --reps 4 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
unset MISO_FLAT_NO_UNI
for lib in miso_amd/libmiso_amd.so tools/_build/libmiso_grpnolicm.so miso_amd/libmiso_amd.so tools/_build/libmiso_grpnolicm.so; do
  echo "== $lib" >> $O/ab.txt
  MISO_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 900 python tools/r6_clock.py pe_k5 pe_k10 pe_mix --reps 3 --probe 0 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
