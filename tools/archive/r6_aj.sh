#!/bin/bash
# round 6: (a) the two-isoform units without machine-level hoisting (variant library) against the build; (b) sampler_flat's chains per
# wavefront and threshold skipping re-swept on the faster kernel
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aj; mkdir -p $O
for lib in miso_amd/libmiso_amd.so tools/_build/libmiso_k2nolicm.so miso_amd/libmiso_amd.so tools/_build/libmiso_k2nolicm.so; do
  echo "== $lib" >> $O/k2.txt
  MISO_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 900 python tools/r6_clock.py main se_k2_hg19 se_k2_defaults se_k2_hg19_defaults pe_k2 pe_k2_hg19 --reps 3 2>&1 | grep -E "kernels|median" | cut -c1-150 >> $O/k2.txt
done
cat $O/k2.txt
MISO_AMD_LIB=$GRAFT_REPO_ROOT/tools/_build/libmiso_k2nolicm.so timeout 900 python -m pytest tests -m gpu -x -q -k "k2 or two_isoform or parity or fuzz or collapsed or paired" > $O/tests.log 2>&1; echo "variant tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
run() { echo "== $*" >> $O/flat.txt; env "$@" timeout 600 python tools/r6_clock.py $ROWS --reps 2 2>&1 | grep -E "median" | cut -c1-60 >> $O/flat.txt; }
ROWS="se_k5"; for nc in 6 7 8 9 10 12; do run MISO_FLAT_NC=$nc; done
run MISO_FLAT_THR_SKIP=0; run MISO_FLAT_THR_SKIP=1
ROWS="se_k10"; for nc in 4 5 6 7 8; do run MISO_FLAT_NC=$nc; done
run MISO_FLAT_THR_SKIP=0; run MISO_FLAT_THR_SKIP=1
ROWS="se_k5_hg19"; for f in 0.7 0.8 0.85 0.9 1.0; do run MISO_FLAT_ROUNDS_FRAC=$f; done
run MISO_FLAT_THR_SKIP=0; run MISO_FLAT_THR_SKIP=1
cat $O/flat.txt
