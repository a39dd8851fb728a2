#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ap; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v 'no chains' $O/tests.log | tail -1)"
grep -v "no chains" $O/tests.log | grep -E "^E |FAILED" | head -20
