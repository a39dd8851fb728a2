#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summary + PMC passes for bench.py.
# Usage: tools/profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MISO_NO_AUTOTUNE=1   # no trial launches in the kernel statistics (K > 2 only; same lanes per chain as the rule of thumb)
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-matrix $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/pmc_write -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -30
