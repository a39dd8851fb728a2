#!/bin/bash
# round 5: queue priorities for the class kernels of a whole-gene batch (longest class first) against default priorities
mkdir -p gpurun_out/r5
out=gpurun_out/r5/prio_ab.txt; : > $out
run() {  # label bench-args -- env...
  label=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-streams "${args[@]}" --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$label $*', d['value'], d['roofline']['kernel'][:60], d['roofline']['kernel_ms'])" >> $out
}
python - >> $out <<'PY'
import ctypes
h = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
PY
for rep in 1 2; do
  for pr in 1 0; do
    run "PE mix prio=$pr" --K-range 3 20 --paired --events 16384 -- MISO_STREAM_PRIO=$pr
    run "PE mix hg19 prio=$pr" --K-range 3 20 --paired --events 16384 --reads-dist hg19 -- MISO_STREAM_PRIO=$pr
  done
done
run "SE mix prio=1" --K-range 3 20 --events 16384 -- MISO_STREAM_PRIO=1
run "SE mix prio=0" --K-range 3 20 --events 16384 -- MISO_STREAM_PRIO=0
timeout 900 python -m pytest tests/test_gpu_heavy_tail.py tests/test_gpu_parity.py -x -q > gpurun_out/r5/prio_tests.log 2>&1
echo "heavy-tail + parity tests rc=$? $(tail -1 gpurun_out/r5/prio_tests.log)" >> $out
cat $out
