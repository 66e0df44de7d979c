#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summaries land in gpurun_out/<tag>/
# usage (on the GPU box, via gpurun): bash tools/prof_kernel_trace.sh <tag> [extra bench args]
tag=${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag -o bench -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag/bench.log 2>&1
tail -1 gpurun_out/$tag/bench.log | cut -c1-400
find gpurun_out/$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -12
# the full trace is large; keep only the stats summaries
find gpurun_out/$tag -name '*kernel_trace.csv' -delete
