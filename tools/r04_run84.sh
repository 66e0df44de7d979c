#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run84; mkdir -p $o
timeout 300 python3 -u -m pytest tests/test_cross_attn_sliced_bwd_gpu.py tests/test_train_chains_gpu.py tests/test_timed_size_parity_gpu.py tests/test_training_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
for v in dot own; do
GD4D_TRAIN_HEADS_BWD=$v timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout > $o/t_${v}_$rep.json 2> $o/t_${v}_$rep.err; echo "heads_bwd=$v $(ms $o/t_${v}_$rep.json)"
done
done
