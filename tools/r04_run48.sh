#!/bin/bash
# the training lines again on the final build (attention backward changed after the bundle)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04b/train; mkdir -p $o gpurun_out/r04b
python3 bench.py --mode train --steps 20 --warmup 3 > $o/train.json 2> $o/train.err
python3 bench.py --mode train --steps 20 --warmup 3 --dropout > $o/train_dropout.json 2> $o/train_dropout.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline > $o/train_generic.json 2> $o/train_generic.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline --dropout > $o/train_generic_dropout.json 2> $o/train_generic_dropout.err
python3 bench.py --mode train --steps 10 --warmup 3 --criterion > $o/train_criterion.json 2> $o/train_criterion.err
python3 bench.py --mode train --steps 5 --warmup 2 --levels vov > $o/train_vov.json 2> $o/train_vov.err
python3 bench.py --mode distill --steps 5 --warmup 2 > $o/distill.json 2> $o/distill.err
bash tools/prof_train_stats.sh r04b/train --no-roofline | head -3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 4 --warmup 2 --no-roofline --dropout > $o/train_tl.json 2> $o/train_tl.err
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train_dropout.txt
find $o/tl -name '*kernel_trace.csv' -delete
for f in train train_dropout train_generic train_generic_dropout train_criterion train_vov distill; do echo "$f $(tail -1 $o/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))')"; done
ulimit -c 0
timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04b/pytest_final.log 2>&1; echo "all rc=$? $(tail -1 gpurun_out/r04b/pytest_final.log)"
