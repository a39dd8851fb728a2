#!/bin/bash
# round 5: (a) sampler_grp / sampler_flat with the iteration's sample stored behind the Gibbs step (variant library) against
# before it (in-tree objects); (b) where the wavefronts of the K >= 3 kernels wait: SQ wait counters per kernel
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
out=gpurun_out/r5/waits.txt; : > $out
V=tools/_build/libmiso_gf_storeafter.so
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
if [ -f $V ]; then
  MISO_AMD_LIB=$V timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_heavy_tail.py tests/test_gpu_paired_dense.py -x -q > gpurun_out/r5/gf_storeafter_tests.log 2>&1
  echo "store-after variant (grp, flat): tests rc=$? $(tail -1 gpurun_out/r5/gf_storeafter_tests.log)" >> $out
  for cfg in "--K 5" "--K 10" "--K 3" "--K 5 --paired" "--K 10 --paired --events 20000" "--K 3 --paired" "--K-range 3 20 --paired --events 16384" "--K 5 --reads-dist hg19"; do
    run "before $cfg" $cfg -- MISO_X=0
    run "after  $cfg" $cfg -- MISO_AMD_LIB=$V
  done
fi
cat $out
pmc() {  # tag bench-args
  tag=$1; shift
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/r5/pmc_$tag -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-matrix --no-streams "$@" > gpurun_out/r5/pmc_$tag.json 2> gpurun_out/r5/pmc_$tag.log
  python3 - <<PY >> gpurun_out/r5/waits.txt
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
pmc pe_k2 --K 2 --paired
pmc se_k2 --K 2
tail -12 $out
