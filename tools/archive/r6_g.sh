#!/bin/bash
# round 6, seventh GPU call: priority by progress (MISO_PRIO_QUARTILES=1, device.hpp prio_by_progress) on the rows whose
# launches take several rounds of workgroups, A/B on one box.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired_dense.py -m gpu -q -x 2>&1 | tail -2
for q in 0 1 0 1; do
  echo "== MISO_PRIO_QUARTILES=$q" >> $O/prio.txt
  MISO_PRIO_QUARTILES=$q timeout 900 python tools/r6_clock.py se_k2_defaults se_k2_hg19_defaults pe_k2 se_k5 se_k10 se_k5_hg19 pe_k5 pe_k10 pe_mix --reps 3 --probe 0 2>&1 | grep -E "kernels|median" | paste - - | sed 's/  kernels.*  median/ median/' >> $O/prio.txt
done
cat $O/prio.txt
