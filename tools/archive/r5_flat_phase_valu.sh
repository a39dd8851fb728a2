#!/bin/bash
# GPU: VALU wave-instructions of sampler_flat by phase -- the full kernel, without the read loop, and without
# read loop and thresholds (libraries built with -DMISO_FLAT_SKIP_LOOP / -DMISO_FLAT_SKIP_THR by
# `EXTRA=... bash tools/build_prof.sh`; results of those builds are wrong, only the counts matter).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
out=gpurun_out/r5/flat_phase_valu.txt; : > $out
for lib in full noloop noloop_nothr; do
  L=$PWD/tools/_build/libmiso_$lib.so; [ $lib = full ] && L=$PWD/miso_amd/libmiso_amd.so
  for K in 5 10; do
    d=gpurun_out/pv_${lib}_$K; rm -rf $d; mkdir -p $d
    MISO_NO_AUTOTUNE=1 MISO_AMD_LIB=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $d -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-matrix --K $K --iters 1500 --burn 500 > $d/bench.json 2> $d/log.txt
    python3 - "$d" "$lib" "$K" >> $out <<'PY'
import glob, sqlite3, sys
d, lib, K = sys.argv[1:4]
db = glob.glob(d + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%sampler_%' group by kernel_name, counter_name").fetchall()
v = {c: x for k, c, x in rows}
print("K=%s %-13s %s: VALU per chain-iteration %.1f (waves %d)" % (K, lib, rows[0][0][:40], v["SQ_INSTS_VALU"] / (40000 * 1501.0), v["SQ_WAVES"]))
PY
    rm -rf $d
  done
done
cat $out
