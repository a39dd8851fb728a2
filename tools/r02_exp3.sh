#!/bin/bash
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest3a.log 2>&1; tail -3 gpurun_out/r02/pytest3a.log
MISO_FLAT_NC=7 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest3b.log 2>&1; tail -3 gpurun_out/r02/pytest3b.log
MISO_AMD_LIB=tools/_build/libmiso_prof.so KS=${KS:-3,5,10} NCS=${NCS:-0} python tools/phase_prof_flat.py 2>&1 | tee gpurun_out/r02/flat_phase_last.txt
