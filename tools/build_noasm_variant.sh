#!/bin/bash
# a build WITHOUT the loads the compiler does not see (pe_dense's asm records at seven / eight isoforms, flat_units_desc's
# asm descriptor prefetch), to be run through the bit-exact GPU tests beside the product build at every compiler bump:
#   bash tools/build_noasm_variant.sh && MISO_AMD_LIB=tools/_build/libmiso_noasm.so python -m pytest tests -m gpu -q \
#        tests/test_gpu_parity.py tests/test_gpu_paired_dense.py tests/test_gpu_fuzz.py tests/test_gpu_heavy_tail.py
cd "$(dirname "$0")/.."
exec bash tools/build_variant.sh noasm "-DMISO_PE_ASM_RECORDS=0 -DMISO_FLAT_ASM_PREFETCH=0" kernels_grp_c8 kernels_grp_all kernels_flat_c4 kernels_flat_c8 kernels_flat_c12 kernels_flat_c16 kernels_flat_c32
