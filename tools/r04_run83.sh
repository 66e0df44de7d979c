#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run83; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
for v in 0 1; do
GD4D_REFPOINTS_TORCH=$v timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout > $o/n_${v}_$rep.json 2> $o/n_${v}_$rep.err; echo "torch=$v $(ms $o/n_${v}_$rep.json)"
done
done
