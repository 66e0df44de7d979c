#!/bin/bash
# head position-embedding backward on the library's kernels: unit tests, then a timing of forward+backward both routes at 24 cameras
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run20; mkdir -p $o
timeout 900 python3 -m pytest tests/test_head_pe_gpu.py tests/test_abi.py -x -q -m gpu -p no:cacheprovider > $o/pytest_pe.log 2>&1; echo "pe rc=$? $(tail -3 $o/pytest_pe.log)"
timeout 600 python3 tools/time_head_pe_train.py > $o/time.txt 2>&1; cat $o/time.txt | tail -20
