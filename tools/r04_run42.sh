#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run42; mkdir -p $o
for ch in 1 0; do
GD4D_TRAIN_CHAINS=$ch python3 bench.py --mode train --queries 2700 --steps 10 --warmup 3 --no-roofline --dropout > $o/q2700_$ch.json 2> $o/q2700_$ch.err
echo "2700 queries chains=$ch $(tail -1 $o/q2700_$ch.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"
GD4D_TRAIN_CHAINS=$ch python3 bench.py --mode train --frames 1 --steps 10 --warmup 3 --no-roofline --dropout > $o/f1_$ch.json 2> $o/f1_$ch.err
echo "6 cameras chains=$ch $(tail -1 $o/f1_$ch.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"
done
