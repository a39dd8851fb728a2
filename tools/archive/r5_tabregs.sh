#!/bin/bash
# round 5: (a) the two-isoform kernels with the exp / log coefficients in VGPRs (variant library) against scalar loads at
# every call; (b) SQ counters of sampler_lane_ilp and sampler_lane
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
out=gpurun_out/r5/tabregs.txt; : > $out
V=tools/_build/libmiso_k2_tabregs.so
MISO_AMD_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -k "single_end or two_isoform or golden_inputs or threshold" > gpurun_out/r5/tabregs_tests.log 2>&1
echo "parity tests on the variant rc=$? $(tail -1 gpurun_out/r5/tabregs_tests.log)" >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for rep in 1 2 3; do
  run "headline in-tree" --K 2 -- MISO_X=0
  run "headline tabregs" --K 2 -- MISO_AMD_LIB=$V
done
run "defaults in-tree" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "defaults tabregs" --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_AMD_LIB=$V
run "hg19 in-tree" --reads-dist hg19 -- MISO_X=0
run "hg19 tabregs" --reads-dist hg19 -- MISO_AMD_LIB=$V
run "hg19 defaults in-tree" --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_X=0
run "hg19 defaults tabregs" --reads-dist hg19 --chains 6 --iters 5000 --burn 500 --lag 10 -- MISO_AMD_LIB=$V
cat $out
for form in 1 0; do
  export MISO_LANE_ILP=$form
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d gpurun_out/r5/pmc_lane_$form -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matrix --no-streams --collapsed 1 > gpurun_out/r5/pmc_lane_$form.json 2> gpurun_out/r5/pmc_lane_$form.log
  python3 - <<PY >> gpurun_out/r5/tabregs.txt
import glob, sqlite3
for db in glob.glob("gpurun_out/r5/pmc_lane_$form/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%sampler_lane%' group by kernel_name, counter_name"):
        print("MISO_LANE_ILP=$form", r[0][:40], r[1], r[2])
PY
  rm -rf gpurun_out/r5/pmc_lane_$form
done
unset MISO_LANE_ILP
tail -20 $out
