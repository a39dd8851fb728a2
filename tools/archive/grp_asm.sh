#!/bin/bash
# (CPU) assembly of the sampler_grp classes -> /tmp/g<KC>.s, register / spill summary
cd "$(dirname "$0")/.."
for c in ${CLASSES:-4 8 12 16 32}; do
  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iinclude -Imiso_amd/csrc --cuda-device-only ${EXTRA} -S -o /tmp/g$c.s miso_amd/csrc/kernels_grp_c$c.hip 2>/dev/null &
done; wait
python - <<PY
import re
for c in "${CLASSES:-4 8 12 16 32}".split():
    t=open('/tmp/g%s.s'%c).read()
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
        n=m.group(1)
        if 'Lb1E' in n: print(c, n[22:40], 'vgpr', m.group(2), 'spill', m.group(3), 'flat ops', 0)
    print(c, 'flat ops in file', len(re.findall(r'\n\s+flat_(load|store|atomic)', t)))
PY
