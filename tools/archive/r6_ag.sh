#!/bin/bash
# round 6: sampler_flat compiled without machine-level hoisting (Makefile FLAT_UNITFLAGS) against the same source with it
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ag; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "flat or parity or fuzz or heavy or wide or convergent" > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for lib in miso_amd/libmiso_amd.so tools/_build/libmiso_licm.so miso_amd/libmiso_amd.so tools/_build/libmiso_licm.so; do
  echo "== $lib" >> $O/ab.txt
  MISO_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 900 python tools/r6_clock.py se_k5 se_k10 se_k5_hg19 --reps 4 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/ab.txt
done
cat $O/ab.txt
