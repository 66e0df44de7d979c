#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/traintl
rocprofv3 --kernel-trace -f csv -d gpurun_out/traintl -o tr -- python3 bench.py --mode train --steps 5 --warmup 2 > gpurun_out/traintl/line.json 2> gpurun_out/traintl/err.txt
t=$(find gpurun_out/traintl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > gpurun_out/traintl/timeline.txt
find gpurun_out/traintl -name '*kernel_trace.csv' -delete
find gpurun_out -name '*agent_info.csv' -delete
head -5 gpurun_out/traintl/timeline.txt
