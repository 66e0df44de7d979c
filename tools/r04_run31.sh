#!/bin/bash
# the round's measurement bundle again (the kernel sources changed: PMC records are keyed to their hashes) + the training lines
cd "$GRAFT_REPO_ROOT"
bash tools/prof_round.sh r04b > gpurun_out/r04b_round.log 2>&1
o=gpurun_out/r04b/train
python3 bench.py --mode train --steps 20 --warmup 3 --dropout > $o/train_dropout.json 2> $o/train_dropout.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline > $o/train_generic.json 2> $o/train_generic.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline --dropout > $o/train_generic_dropout.json 2> $o/train_generic_dropout.err
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 4 --warmup 2 --no-roofline --dropout > $o/train_tl.json 2> $o/train_tl.err
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train_dropout.txt
find $o/tl -name '*kernel_trace.csv' -delete
for f in train train_dropout train_generic train_generic_dropout; do echo "$f $(tail -1 $o/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("query_side","")[:40])')"; done
tail -5 gpurun_out/r04b_round.log | cut -c1-200
