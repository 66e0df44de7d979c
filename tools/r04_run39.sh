#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run39; mkdir -p $o
for rep in 1 2; do for ch in 1 0; do
GD4D_TRAIN_CHAINS=$ch python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline --no-graph > $o/eager_$ch_$rep.json 2> $o/eager_$ch_$rep.err
echo "eager chains=$ch $(tail -1 $o/eager_$ch_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["launch"][:40])')"
done; done
