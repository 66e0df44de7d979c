#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04c
for rep in 1 2; do
python3 bench.py --mode train --steps 30 --warmup 3 > gpurun_out/r04c/train_$rep.json 2> gpurun_out/r04c/train.err
python3 bench.py --mode train --steps 30 --warmup 3 --dropout > gpurun_out/r04c/train_dropout_$rep.json 2> gpurun_out/r04c/train_dropout.err
for f in train_$rep train_dropout_$rep; do tail -1 gpurun_out/r04c/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["traffic"])'; done
done
