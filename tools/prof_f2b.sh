#!/bin/bash
# rocprofv3 kernel statistics of the features -> boxes request (bench.py --only-f2b): profiles/r06_kernel_stats_features_to_boxes.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/f2b
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/f2b -o f2b -- python3 bench.py --only-f2b --steps 10 > gpurun_out/f2b/bench.log 2>&1
tail -1 gpurun_out/f2b/bench.log | cut -c1-600
find gpurun_out/f2b -name '*kernel_trace.csv' -delete
find gpurun_out/f2b -name '*kernel_stats.csv' | head -1 | xargs -r head -30
