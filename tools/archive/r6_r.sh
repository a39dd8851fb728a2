#!/bin/bash
# round 6: per-class kernel durations of the whole-gene mix under the CU partition (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6r; mkdir -p $O
export MISO_CLASS_CUMASK=1 MISO_TIMING=1
for row in pe_mix pe_mix_hg19; do
  rocprofv3 --kernel-trace --stats -d $O/tr_$row -o t -- python3 tools/r6_clock.py $row --reps 3 --probe 0 > $O/$row.log 2>&1
  python3 - <<EOF
import glob, sqlite3
db = glob.glob("$O/tr_$row/*.db")[0]
con = sqlite3.connect(db)
print("$row")
for r in con.execute("select name, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 from kernels where name like '%sampler_%' group by name order by 3 desc"):
    print("  %-62s n %d avg %.1f ms min %.1f max %.1f" % (r[0][:62], r[1], r[2], r[3], r[4]))
EOF
  grep -E "class [0-9]+: CUs" $O/$row.log | sort | uniq
  rm -rf $O/tr_$row
done
