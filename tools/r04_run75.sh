#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run75; mkdir -p $o
for f in test_train_chains_gpu test_training_gpu test_timed_size_parity_gpu test_dense_gpu; do
timeout 240 python3 -u -m pytest tests/$f.py -x -v -m gpu -p no:cacheprovider > $o/$f.log 2>&1; echo "$f rc=$? $(tail -1 $o/$f.log)"; tail -4 $o/$f.log | head -3 | cut -c1-200
done
