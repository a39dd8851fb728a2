#!/bin/bash
# Variant library for A/B runs: recompile only the listed translation units with extra defines, link with
# the in-tree objects of everything else.   tools/build_variant.sh NAME "-DMISO_X=1" kernels_grp_c8 kernels_grp_c12
# -> tools/_build/libmiso_NAME.so   (run with MISO_AMD_LIB=tools/_build/libmiso_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2; shift 2
mkdir -p tools/_build/$name
cp miso_amd/csrc/*.o tools/_build/$name/
F="$defs -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Imiso_amd/csrc -Wno-unused-result"
# (the flat and grp units carry the Makefile's own extra flags; FLAT_UNITFLAGS="" builds them without)
FU=${FLAT_UNITFLAGS--mllvm -disable-machine-licm}
for f in "$@"; do U=""; case $f in kernels_flat_c*|kernels_grp_c*|kernels_grp_all|kernels_k2m_m0w4n) U=$FU;; esac; /opt/rocm/bin/hipcc $F $U -c miso_amd/csrc/$f.hip -o tools/_build/$name/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/_build/$name/*.o -o tools/_build/libmiso_$name.so -lz -lpthread
rm -rf tools/_build/$name
