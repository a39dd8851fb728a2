#!/bin/bash
# round 5: pe_dense's score gathers a trip late (MISO_PE_LATE_GATHER) and sampler_flat's descriptor prefetch that is not
# waited for at once (MISO_FLAT_ASM_PREFETCH): parity, then the rows
mkdir -p gpurun_out/r5
out=gpurun_out/r5/prefetch.txt; : > $out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/prefetch_tests.log 2>&1
echo "gpu tests rc=$? $(grep -v 'no chains' gpurun_out/r5/prefetch_tests.log | tail -1)" >> $out
grep -E "^E|FAILED" gpurun_out/r5/prefetch_tests.log | head -10 >> $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:90], d['roofline']['kernel_ms'])" >> $out
}
for cfg in "--K 3" "--K 4" "--K 5" "--K 6" "--K 8" "--K 10" "--K 12" "--K 16" "--K 5 --reads-dist hg19" "--K-range 3 20 --events 16384" "--K 6 --paired" "--K 7 --paired" \
           "--K 3 --paired" "--K 5 --paired" "--K 8 --paired" "--K 10 --paired --events 20000" "--K 16 --paired --events 20000" \
           "--K 5 --paired --reads-dist hg19" "--K-range 3 20 --paired --events 16384" "--K-range 3 20 --paired --events 16384 --reads-dist hg19"; do
  run "now $cfg" $cfg -- MISO_X=0
done
cat $out
