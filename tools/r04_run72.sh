#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04c
python3 bench.py > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err
tail -1 gpurun_out/r04c/bench.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('default:', round(d['value'],1), d['ms_per_step'], 'batch1', d.get('value_batch1'), 'frac', r['frac'], 'traffic', r['traffic'])"
python3 bench.py --mode train --steps 20 --warmup 3 > gpurun_out/r04c/train.json 2> gpurun_out/r04c/train.err
python3 bench.py --mode train --steps 20 --warmup 3 --dropout > gpurun_out/r04c/train_dropout.json 2> gpurun_out/r04c/train_dropout.err
for f in train train_dropout; do tail -1 gpurun_out/r04c/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["traffic"])'; done
