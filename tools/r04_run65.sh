#!/bin/bash
# what-if: the attention backward with its second products' operands as coalesced fragments (wrong results; timing only)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
echo "base:   $(python3 tools/bench_train_small.py 2>/dev/null | grep 'mha core' | tr '\n' ';')"
echo "whatif: $(GD4D_LIB_PATH=$PWD/build_ab/libgd4d_whatif.so python3 tools/bench_train_small.py 2>/dev/null | grep 'mha core' | tr '\n' ';')"
done
