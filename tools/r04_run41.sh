#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run41; mkdir -p $o
timeout 900 python3 -m pytest tests/test_end_to_end_gpu.py -x -q -m gpu -p no:cacheprovider --durations=3 > $o/e2e.log 2>&1; echo "rc=$? $(tail -1 $o/e2e.log)"; grep -n "^E \|slowest\|s call" $o/e2e.log | head -8
