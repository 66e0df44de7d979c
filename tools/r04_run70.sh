#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run70; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3 4 5 6; do
for v in prepare 0; do
GD4D_TRAIN_SIDE=$v python3 bench.py --mode train --steps 60 --warmup 5 --no-roofline --dropout > $o/t_${v}_$rep.json 2> $o/t_${v}_$rep.err; echo "train side=$v $(ms $o/t_${v}_$rep.json)"
done
done
