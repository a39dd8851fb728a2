#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r6s; mkdir -p $O
export MISO_TIMING=1
run() {  # env assignments
  echo "== $*" >> $O/cumask2.txt
  env "$@" rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/r6_clock.py pe_mix --reps 3 --probe 0 > $O/run.log 2>&1
  grep -E "median" $O/run.log >> $O/cumask2.txt
  python3 - >> $O/cumask2.txt <<EOF
import glob, sqlite3
db = glob.glob("$O/tr/*.db")[0]
con = sqlite3.connect(db)
for r in con.execute("select name, count(*), avg(end-start)/1e6 from kernels where name like '%sampler_%' group by name order by 3 desc"):
    print("  %-62s n %d avg %.1f ms" % (r[0][:62], r[1], r[2]))
EOF
  grep -E "class [0-9]+: CUs" $O/run.log | sort | uniq >> $O/cumask2.txt
  rm -rf $O/tr
}
S="4:0.036,8:0.107,12:0.223,16:0.245,32:0.388"
run MISO_CLASS_CUMASK=1 MISO_CLASS_SHARE=$S
run MISO_CLASS_CUMASK=2
run MISO_CLASS_CUMASK=2 MISO_CLASS_SHARE=$S
cat $O/cumask2.txt
