#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run50; mkdir -p $o
for rep in 1 2 3; do for v in 1 0; do
GD4D_TRAIN_DEFER_VP=$v python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/vp${v}_$rep.json 2> $o/vp${v}_$rep.err; echo "defer_vp=$v $(tail -1 $o/vp${v}_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
