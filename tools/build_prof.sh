#!/bin/bash
# Profile build of the library (-DMISO_K2_PROFILE: s_memtime phase counters written into loglik[0..]).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
F="$EXTRA ${PROFDEF--DMISO_K2_PROFILE} -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Imiso_amd/csrc -Wno-unused-result"
for f in runtime kernels kernels_k2 kernels_grp_c4 kernels_grp_c8 kernels_grp_c12 kernels_grp_c16 kernels_grp_c32 kernels_flat_c4 kernels_flat_c8 kernels_flat_c12 kernels_flat_c16 kernels_flat_c32 kernels_summary kernels_match capi; do
  /opt/rocm/bin/hipcc $F -c miso_amd/csrc/$f.hip -o tools/_build/$f.o &
done
/opt/rocm/bin/hipcc $F -x hip -c miso_amd/csrc/host.cpp -o tools/_build/host.o &
g++ -O2 -std=c++17 -fPIC -Iinclude -Imiso_amd/csrc -c miso_amd/csrc/alnio.cpp -o tools/_build/alnio.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/_build/*.o -o tools/_build/${OUTLIB:-libmiso_prof.so} -lz -lpthread
rm -f tools/_build/*.o
