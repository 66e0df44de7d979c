#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run5; mkdir -p $o
timeout 900 python3 -m pytest tests/test_cross_attn_sliced_gpu.py tests/test_head_pe_gpu.py tests/test_abi.py -x -q -m gpu --durations=5 > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $o/pytest.log
