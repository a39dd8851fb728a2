#!/bin/bash
# round 6, tenth GPU call: the bit-exact tests on the build WITHOUT the hidden loads, then the whole suite on the product.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
MISO_AMD_LIB=tools/_build/libmiso_noasm.so timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py tests/test_gpu_heavy_tail.py > $O/noasm_tests.log 2>&1
echo "no-asm variant: rc=$? $(grep -E 'passed|failed' $O/noasm_tests.log | tail -1)"
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/gputests.log | tail -1)"
grep -E "^E|FAILED" $O/gputests.log | head -10
