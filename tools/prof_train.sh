#!/bin/bash
# Training-step measurement bundle (run on the GPU box through gpurun):
#   1. bench lines: synthetic loss (one hipGraph), head + Hungarian loss (--criterion), VoVNet-size pyramid
#   2. rocprofv3 kernel stats of the eager training step -> gpurun_out/<tag>/stats/  (copy to profiles/ to keep)
tag=${1:-train}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
python3 bench.py --mode train --steps 10 --warmup 3 > gpurun_out/$tag/train.json 2> gpurun_out/$tag/train.err
python3 bench.py --mode train --steps 10 --warmup 3 --criterion > gpurun_out/$tag/train_criterion.json 2> gpurun_out/$tag/train_criterion.err
python3 bench.py --mode train --steps 5 --warmup 2 --levels vov > gpurun_out/$tag/train_vov.json 2> gpurun_out/$tag/train_vov.err
for f in train train_criterion train_vov; do tail -1 gpurun_out/$tag/$f.json | cut -c1-200; done
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/stats -o train -- python3 bench.py --mode train --steps 5 --warmup 2 --no-graph > gpurun_out/$tag/profiled.json 2> gpurun_out/$tag/profiled.err
find gpurun_out/$tag/stats -name '*kernel_trace.csv' -delete
head -12 gpurun_out/$tag/stats/train_kernel_stats.csv | cut -c1-160
