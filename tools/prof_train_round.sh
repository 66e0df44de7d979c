#!/bin/bash
# Training part of the round bundle alone (bench lines of the step's variants + kernel stats + PMC of the raw-pyramid backward)
tag=${1:-train_round}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag/train
python3 bench.py --mode train --steps 20 --warmup 3 > gpurun_out/$tag/train/train.json 2> gpurun_out/$tag/train/train.err
python3 bench.py --mode train --steps 10 --warmup 3 --criterion > gpurun_out/$tag/train/train_criterion.json 2> gpurun_out/$tag/train/train_criterion.err
python3 bench.py --mode train --steps 5 --warmup 2 --levels vov > gpurun_out/$tag/train/train_vov.json 2> gpurun_out/$tag/train/train_vov.err
python3 bench.py --mode distill --steps 5 --warmup 2 > gpurun_out/$tag/train/distill.json 2> gpurun_out/$tag/train/distill.err
GD4D_TRAIN_VALUES=projected python3 bench.py --mode train --steps 10 --warmup 3 --no-fuse-wgrad > gpurun_out/$tag/train/train_projected_values.json 2> gpurun_out/$tag/train/train_projected_values.err
for f in train train_criterion train_vov distill train_projected_values; do tail -1 gpurun_out/$tag/train/$f.json | cut -c1-200; done
bash tools/prof_train_stats.sh $tag/train --no-roofline | head -40
