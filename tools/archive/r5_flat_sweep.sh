#!/bin/bash
# round 5: chains per wavefront of sampler_flat again, after the descriptor prefetch stopped waiting (the rule's numbers are round 3's)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_sweep.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:40], d['roofline']['kernel_ms'])" >> $out
}
# collapsed mode: fewer than 64 chains per wavefront when the batch has fewer wavefronts than the device has SIMDs
timeout 600 python -m pytest tests/test_gpu_collapsed.py tests/test_gpu_convergent.py -x -q > gpurun_out/r5/spread_tests.log 2>&1
echo "collapsed tests rc=$? $(tail -1 gpurun_out/r5/spread_tests.log)" >> $out
for rep in 1 2; do
  run "spread" --collapsed 1 -- MISO_X=0
  run "64 per wavefront" --collapsed 1 -- MISO_LANE_SPREAD=0
done
run "hg19 spread" --collapsed 1 --reads-dist hg19 -- MISO_X=0
run "hg19 64" --collapsed 1 --reads-dist hg19 -- MISO_LANE_SPREAD=0
run "reads=3000 spread" --collapsed 1 --reads 3000 -- MISO_X=0
run "reads=3000 64" --collapsed 1 --reads 3000 -- MISO_LANE_SPREAD=0
run "20000 events spread" --collapsed 1 --events 20000 -- MISO_X=0
run "20000 events 64" --collapsed 1 --events 20000 -- MISO_LANE_SPREAD=0
run "100000 events spread" --collapsed 1 --events 100000 -- MISO_X=0
run "100000 events 64" --collapsed 1 --events 100000 -- MISO_LANE_SPREAD=0
for K in 3 4 5 6 7 8; do
  run "K=$K rule" --K $K -- MISO_X=0
  for nc in 5 6 7 8 9 10 12 14 16; do run "K=$K nc=$nc" --K $K -- MISO_FLAT_NC=$nc; done
done
for K in 9 10 12; do
  run "K=$K rule" --K $K -- MISO_X=0
  for nc in 3 4 5 6 7 8; do run "K=$K nc=$nc" --K $K -- MISO_FLAT_NC=$nc; done
done
for K in 14 16 20; do
  run "K=$K rule" --K $K --events 20000 -- MISO_X=0
  for nc in 2 3 4 5 6; do run "K=$K nc=$nc" --K $K --events 20000 -- MISO_FLAT_NC=$nc; done
done
cat $out
# the wait counters of profiles/r05_wait_counters.txt again, on the final kernels
: > gpurun_out/r5/waits_after.txt
pmc() {  # tag bench-args
  tag=$1; shift
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/r5/pmc_$tag -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matrix --no-streams "$@" > gpurun_out/r5/pmc_$tag.json 2> gpurun_out/r5/pmc_$tag.log
  python3 - <<PY >> gpurun_out/r5/waits_after.txt
import glob, sqlite3
for db in glob.glob("gpurun_out/r5/pmc_$tag/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    rows = {}
    for k, c, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%sampler_%' group by kernel_name, counter_name"):
        rows.setdefault(k, {})[c] = v
    for k, r in rows.items():
        w = r.get("SQ_WAVE_CYCLES", 0) or 1
        print("$tag", k[6:60], " ".join("%s=%.3f" % (c[3:], r[c] / w) for c in sorted(r) if c != "SQ_WAVE_CYCLES" and c != "SQ_INSTS_VALU"), "valu_per_wavecycle4=%.3f" % (r.get("SQ_INSTS_VALU", 0) / w))
PY
  rm -rf gpurun_out/r5/pmc_$tag
}
pmc pe_k5 --K 5 --paired
pmc pe_k10 --K 10 --paired --events 20000
pmc se_k5 --K 5
pmc se_k10 --K 10
cat gpurun_out/r5/waits_after.txt
