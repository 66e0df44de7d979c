#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run79; mkdir -p $o
ulimit -c 0
timeout 420 python3 -u -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/all.log 2>&1; echo "all rc=$? $(tail -1 $o/all.log)"; grep -n "^E " $o/all.log | head -6
