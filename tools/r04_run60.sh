#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run60; mkdir -p $o
GD4D_CHECK_HANDOFF=1 timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
