#!/bin/bash
# round 5, run 2: the merging walk - parity + A/B timing
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_run2; mkdir -p $o
timeout 900 python3 -m pytest tests/test_cross_attn_sliced_gpu.py -x -q -m gpu > $o/pytest_sliced.log 2>&1; echo "pytest sliced rc=$?"; tail -15 $o/pytest_sliced.log
for m in 0 2 3 1 0 2; do timeout 300 python3 tools/bench_sliced.py --merge $m 2>&1 | tail -2; done | tee $o/ab.txt
timeout 300 python3 tools/bench_sliced.py --merge 2 --layout pixel 2>&1 | tail -2 | tee -a $o/ab.txt
timeout 300 python3 tools/bench_sliced.py --merge 2 --alias 2>&1 | tail -2 | tee -a $o/ab.txt
