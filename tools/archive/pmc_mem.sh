#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <bench args>   vector-memory-path counters of one launch (TA / vector L1 / L2), own pass
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcm_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $OUT/pmc_m -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matrix "$@" > $OUT/bench.json 2> $OUT/log_m.txt
python3 tools/prof_summary.py $OUT $OUT/summary.txt | grep -E "sampler"
tail -5 $OUT/log_m.txt | grep -i "error\|invalid\|not" | head -5
