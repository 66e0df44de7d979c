#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run47; mkdir -p $o
timeout 900 python3 -m pytest tests/test_dense_gpu.py tests/test_training_gpu.py tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -5
for rep in 1 2; do for v in new:graph-detr4d_amd/libgd4d.so old:build_ab/libgd4d_old.so; do name=${v%%:*}; lib=${v##*:}
echo "$name: $(GD4D_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 tools/bench_train_small.py 2>/dev/null | grep 'mha core' | sed 's/mha core 900 x 900, //' | tr '\n' ';')"
done; done
