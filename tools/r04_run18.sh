#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run18; mkdir -p $o
timeout 900 python3 -m pytest tests/test_training_gpu.py tests/test_abi.py tests/test_modules_gpu.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $o/pytest.log
