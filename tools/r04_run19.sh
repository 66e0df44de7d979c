#!/bin/bash
# flakiness check: the whole GPU suite three times, the hand-off test 20 more
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run19; mkdir -p $o
for i in 1 2 3; do timeout 1200 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/pytest_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $o/pytest_$i.log)"; done
GD4D_SKIP_DP2_DRYRUN=1 timeout 900 python3 -m pytest tests/test_rowchain_gpu.py -q -m gpu -k "signal_wait" --count 1 > /dev/null 2>&1
for i in $(seq 1 10); do GD4D_SKIP_DP2_DRYRUN=1 timeout 300 python3 -m pytest tests/test_rowchain_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "signal_wait or requests_in_flight or schedule or sees_weight" 2>&1 | tail -1; done
