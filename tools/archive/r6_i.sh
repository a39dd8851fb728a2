#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python tools/r6_pe_ws.py 5 10 > $O/pe_working_set.txt 2>&1
cat $O/pe_working_set.txt
