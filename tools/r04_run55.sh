#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04b/train; mkdir -p $o
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 6 --warmup 2 --no-roofline --dropout > $o/train_tl.json 2> $o/train_tl.err
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train_dropout.txt
find $o/tl -name '*kernel_trace.csv' -delete
head -2 $o/timeline_train_dropout.txt
