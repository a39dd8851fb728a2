#!/bin/bash
# round 6: pe_dense as a function of its own (MISO_PE_DENSE_NOINLINE), at two and at three workgroups per CU (variant libraries)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w; mkdir -p $O
for lib in tools/_build/libmiso_pedn2.so tools/_build/libmiso_pedn3.so; do
  echo "== $lib" >> $O/noinline.txt
  MISO_AMD_LIB=$lib timeout 300 python -m pytest -m gpu -q -x tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py 2>&1 | grep -v "no chains" | tail -1 >> $O/noinline.txt
  MISO_AMD_LIB=$lib timeout 600 python tools/r6_clock.py pe_k5 pe_k10 pe_k5_hg19 --reps 3 --probe 0 2>&1 | grep -E "kernels|median" >> $O/noinline.txt
done
echo "== in-tree" >> $O/noinline.txt
timeout 600 python tools/r6_clock.py pe_k5 pe_k10 pe_k5_hg19 --reps 3 --probe 0 2>&1 | grep -E "kernels|median" >> $O/noinline.txt
cat $O/noinline.txt
