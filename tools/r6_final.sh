#!/bin/bash
# round 6 closing run on one GPU box: the whole GPU suite, smoke(), the rocprofv3 passes of every bench row
# (tools/round6_profiles.sh -> profiles/r06_*_summary.txt, valu_model.json, traffic.json), then the default bench line (what
# the driver runs) priced with THOSE profiles, and the end-to-end `miso --run`.   -> gpurun_out/r6z/
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6z/gputests_final.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r6z/gputests_final.log | tail -1)" | tee gpurun_out/r6z/summary.txt
grep -E "^E|FAILED" gpurun_out/r6z/gputests_final.log | head -10 | tee -a gpurun_out/r6z/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r6z/summary.txt
ROUND=06 bash tools/round6_profiles.sh $PROFILE_TAGS > gpurun_out/r6z/profiles.log 2>&1
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null   # (the bench below prices itself with this box's profiles)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6z/bench_default.out 2> gpurun_out/r6z/bench_default.err
echo "bench rc=$?" | tee -a gpurun_out/r6z/summary.txt; tail -c 3990 gpurun_out/r6z/bench_default.out
cp gpurun_out/bench_full.json gpurun_out/r6z/bench_default_full.json
MISO_TIMING=1 timeout 900 python tools/e2e_bench.py --events 40000 --reads 1000 --runs 1:fork,4:fork --summary-only > gpurun_out/r6z/e2e_40000.txt 2>&1
grep -E "^miso --run|^events|Collected|alignment file open" gpurun_out/r6z/e2e_40000.txt
