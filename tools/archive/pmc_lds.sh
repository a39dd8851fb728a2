#!/bin/bash
# usage: tools/pmc_lds.sh <tag> <bench args>   LDS / memory-wait counters of one launch
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcl_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_a -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matrix "$@" > $OUT/bench.json 2> $OUT/log_a.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/pmc_b -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matrix "$@" > $OUT/bench_b.json 2> $OUT/log_b.txt
python3 tools/prof_summary.py $OUT $OUT/summary.txt | grep -E "sampler" 
tail -3 $OUT/log_a.txt $OUT/log_b.txt | grep -i "error\|invalid\|not" | head
