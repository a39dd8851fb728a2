#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest5.log 2>&1; tail -6 gpurun_out/r02/pytest5.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r02/bench_full.json 2> gpurun_out/r02/bench_full.err; echo "bench rc=$?"
tail -3 gpurun_out/r02/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_full.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:900])
print(json.dumps(d['cpu_baseline']))
print(json.dumps(d['delta_psi_vs_reference']))
for r in d.get('matrix', []): print(r)
PY
