#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <bench args>   (one SQ PMC pass, summary printed)
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matrix "$@" > $OUT/bench.json 2> $OUT/log.txt
python3 tools/prof_summary.py $OUT $OUT/summary.txt | grep -E "sampler|kernel"
