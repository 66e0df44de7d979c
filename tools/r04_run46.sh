#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in new:graph-detr4d_amd/libgd4d.so old:build_ab/libgd4d_old.so; do name=${v%%:*}; lib=${v##*:}
echo "$name: $(GD4D_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 tools/bench_train_small.py 2>/dev/null | grep 'mha core' | tr '\n' ';')"
done; done
