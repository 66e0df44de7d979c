#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run21; mkdir -p $o
timeout 900 python3 -m pytest tests/test_head_pe_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "head or gemm or se_fuse or pe" > $o/pytest_pe.log 2>&1; echo "pe rc=$? $(tail -3 $o/pytest_pe.log)"
cd /tmp && export TMPDIR=/tmp
CAMS=24 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$o/prof -o pe -- python3 $GRAFT_REPO_ROOT/tools/time_head_pe_train.py > $GRAFT_REPO_ROOT/$o/prof_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $o/prof -name "*kernel_stats.csv" | head -1); head -25 "$f"
