#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run23; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider -s > $o/pytest_chains.log 2>&1; echo "chains rc=$? $(tail -5 $o/pytest_chains.log)"
