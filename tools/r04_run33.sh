#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run33; mkdir -p $o
timeout 900 python3 -m pytest tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider -s -k "chain_training" > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep "chain vs per-module" $o/tests.log | cut -c1-700; grep -n "^E" $o/tests.log | head -5
