#!/bin/bash
# Profile build of the flat kernels only (-DMISO_K2_PROFILE: phase cycle counters written into loglik[0..3]), linked with
# the product's other objects: tools/_build/libmiso_prof.so.  Run AFTER `make -C miso_amd/csrc`.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
F="$EXTRA -DMISO_K2_PROFILE -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Imiso_amd/csrc -Wno-unused-result"
for kc in ${KCS:-8 12}; do /opt/rocm/bin/hipcc $F -c miso_amd/csrc/kernels_flat_c$kc.hip -o tools/_build/kernels_flat_c$kc.o & done
wait
OBJS=""
for o in miso_amd/csrc/*.o; do b=$(basename $o); if [ -f tools/_build/$b ]; then OBJS="$OBJS tools/_build/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/_build/libmiso_prof.so -lz -lpthread
rm -f tools/_build/*.o
ls -la tools/_build/libmiso_prof.so
