#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run25; mkdir -p $o
for ch in 1 0 1 0; do
GD4D_TRAIN_CHAINS=$ch python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline > $o/train_$ch.json 2> $o/train_$ch.err
echo "chains=$ch $(tail -1 $o/train_$ch.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("value"))')"
done
timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/pytest_all.log 2>&1; echo "all rc=$? $(tail -1 $o/pytest_all.log)"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 4 --warmup 2 --no-roofline > $o/train_prof.json 2> $o/train_prof.err
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train.txt
find $o/tl -name '*kernel_trace.csv' -delete
head -3 $o/timeline_train.txt
