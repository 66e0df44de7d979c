#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run56; mkdir -p $o
GD4D_CHECK_HANDOFF=1 timeout 900 python3 -m pytest tests/test_train_chains_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -5
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
for v in 1 0; do
GD4D_TRAIN_REG_BESIDE=$v python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/b_${v}_$rep.json 2> $o/b_${v}_$rep.err; echo "beside=$v dropout $(ms $o/b_${v}_$rep.json)"
done
done
for v in 1 0; do
GD4D_TRAIN_REG_BESIDE=$v python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline > $o/e_${v}.json 2> $o/e_${v}.err; echo "beside=$v eval $(ms $o/e_${v}.json)"
done
