#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run30; mkdir -p $o
run() { name=$1; shift; env "$@" python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline > $o/$name.json 2> $o/$name.err; echo "$name $(tail -1 $o/$name.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"; }
for rep in 1 2; do
run base_$rep X=1
run count_$rep GD4D_TRAIN_SIDE=count
run wgrad_$rep GD4D_TRAIN_WGRAD_SIDE=1
run copy_$rep GD4D_TRAIN_COPY_SIDE=1
run all_$rep GD4D_TRAIN_SIDE=count GD4D_TRAIN_WGRAD_SIDE=1 GD4D_TRAIN_COPY_SIDE=1
done
timeout 600 python3 -m pytest tests/test_training_gpu.py tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider > $o/t0.log 2>&1; echo "tests default rc=$? $(tail -1 $o/t0.log)"
GD4D_TRAIN_SIDE=count GD4D_TRAIN_WGRAD_SIDE=1 GD4D_TRAIN_COPY_SIDE=1 timeout 600 python3 -m pytest tests/test_training_gpu.py tests/test_train_chains_gpu.py tests/test_configs_gpu.py -x -q -m gpu -p no:cacheprovider > $o/t1.log 2>&1; echo "tests side rc=$? $(tail -1 $o/t1.log)"
