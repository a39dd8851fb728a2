#!/bin/bash
# round 5 closing run on one GPU box: the whole GPU suite, smoke(), the rocprofv3 passes of every bench row
# (tools/round5_profiles.sh -> profiles/r05_*_summary.txt, valu_model.json, traffic.json), then the default bench line (what
# the driver runs) priced with THOSE profiles, and the end-to-end `miso --run`.   -> gpurun_out/r5f/
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/gputests_final.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/r5f/gputests_final.log | tail -1)" | tee gpurun_out/r5f/summary.txt
grep -E "^E|FAILED" gpurun_out/r5f/gputests_final.log | head -10 | tee -a gpurun_out/r5f/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5f/summary.txt
bash tools/round5_profiles.sh > gpurun_out/r5f/profiles.log 2>&1
cp gpurun_out/valu_model.json gpurun_out/traffic.json profiles/ 2>/dev/null   # (the bench below prices itself with this box's profiles)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f/bench_default.out 2> gpurun_out/r5f/bench_default.err
echo "bench rc=$?" | tee -a gpurun_out/r5f/summary.txt; tail -c 3900 gpurun_out/r5f/bench_default.out
cp gpurun_out/bench_full.json gpurun_out/r5f/bench_default_full.json
for ev in 20000 40000; do
  timeout 600 python tools/e2e_bench.py --events $ev --reads 1000 --runs 1:fork,4:fork > gpurun_out/r5f/e2e_$ev.txt 2>&1
  tail -12 gpurun_out/r5f/e2e_$ev.txt
done
