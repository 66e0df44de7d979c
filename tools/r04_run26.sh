#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run26; mkdir -p $o
timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/pytest_all.log 2>&1; echo "all rc=$? $(tail -1 $o/pytest_all.log)"
b() { # name, env..., -- args
  name=$1; shift
  env "$@" python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline $EXTRA > $o/$name.json 2> $o/$name.err
  echo "$name $(tail -1 $o/$name.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"
}
for rep in 1 2; do
EXTRA="" b chains_$rep GD4D_TRAIN_CHAINS=1
EXTRA="" b generic_$rep GD4D_TRAIN_CHAINS=0
EXTRA="" b chains_pairs_$rep GD4D_TRAIN_CHAINS=1 GD4D_TRAIN_PLAN=pairs
EXTRA="" b chains_side_$rep GD4D_TRAIN_CHAINS=1 GD4D_TRAIN_SIDE=1
EXTRA="--dropout" b chains_drop_$rep GD4D_TRAIN_CHAINS=1
EXTRA="--dropout" b generic_drop_$rep GD4D_TRAIN_CHAINS=0
done
