#!/bin/bash
# round 6: the default bench line and the end-to-end run on the final build (profiles of every row: tools/r6_final.sh, r6_k.sh)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6z/gputests_final.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r6z/gputests_final.log | tail -1)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6z/bench_default.out 2> gpurun_out/r6z/bench_default.err
echo "bench rc=$?"; tail -c 3990 gpurun_out/r6z/bench_default.out | cut -c1-700
cp gpurun_out/bench_full.json gpurun_out/r6z/bench_default_full.json
MISO_TIMING=1 timeout 900 python tools/e2e_bench.py --events 40000 --reads 1000 --runs 1:fork --summary-only > gpurun_out/r6z/e2e_40000.txt 2>&1
grep -E "^miso --run|^events|Collected|alignment file open" gpurun_out/r6z/e2e_40000.txt
