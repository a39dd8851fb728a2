#!/bin/bash
# round 6, last: re-profile pe_mix_hg19 (small genes' bucket), the closing run (tools/r6_last.sh), then the parity evidence beyond the suite on this build
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bb; mkdir -p $O
ROUND=06 bash tools/round6_profiles.sh pe_mix_hg19 > $O/profiles.log 2>&1
tail -1 $O/profiles.log | cut -c1-300
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null
bash tools/r6_last.sh
MISO_AMD_LIB=$GRAFT_REPO_ROOT/tools/_build/libmiso_noasm.so timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py tests/test_gpu_heavy_tail.py > $O/noasm_tests.log 2>&1
echo "no-asm variant: rc=$? $(grep -E 'passed|failed' $O/noasm_tests.log | tail -1)"
timeout 1500 python tools/fuzz/gpu_fuzz_convergent.py 10 2>&1 | grep -v "no chains" | tail -2 > $O/fuzz_convergent.txt; cat $O/fuzz_convergent.txt
timeout 2400 python tools/fuzz/gpu_fuzz_more.py 2>&1 | grep -v "no chains" | tail -4 > $O/fuzz_more.txt; cat $O/fuzz_more.txt
