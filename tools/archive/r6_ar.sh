#!/bin/bash
# round 6, late: the parity evidence beyond `pytest -m gpu` again on the final build: the bit-exact tests on the variant without hidden loads,
# the convergent fuzz, the extended fuzz (now with every class on sixteen lanes in one launch: sampler_grp_all)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ar; mkdir -p $O
MISO_AMD_LIB=$GRAFT_REPO_ROOT/tools/_build/libmiso_noasm.so timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py tests/test_gpu_heavy_tail.py > $O/noasm_tests.log 2>&1
echo "no-asm variant: rc=$? $(grep -E 'passed|failed' $O/noasm_tests.log | tail -1)"
timeout 1500 python tools/fuzz/gpu_fuzz_convergent.py 10 2>&1 | grep -v "no chains" | tail -3 > $O/fuzz_convergent.txt; cat $O/fuzz_convergent.txt
timeout 2400 python tools/fuzz/gpu_fuzz_more.py 2>&1 | grep -v "no chains" | tail -5 > $O/fuzz_more.txt; cat $O/fuzz_more.txt
