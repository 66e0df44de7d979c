#!/bin/bash
# kernel trace of the default bench (few steps) + the timeline of one replayed step
tag=${1:-tl}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag -o bench -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > gpurun_out/$tag/bench.log 2>&1
tail -1 gpurun_out/$tag/bench.log | cut -c1-300
t=$(find gpurun_out/$tag -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t $MARKER > gpurun_out/$tag/timeline.txt
find gpurun_out/$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -14 > gpurun_out/$tag/stats_head.txt
find gpurun_out/$tag -name '*kernel_trace.csv' -delete
head -60 gpurun_out/$tag/timeline.txt
