#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run69; mkdir -p $o
GD4D_TRAIN_SIDE=prepare timeout 600 python3 -m pytest tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests(prepare on side) rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
for v in prepare 0; do
GD4D_TRAIN_SIDE=$v python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/t_${v}_$rep.json 2> $o/t_${v}_$rep.err; echo "train side=$v $(ms $o/t_${v}_$rep.json)"
done
done
