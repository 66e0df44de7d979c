#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run78; mkdir -p $o
timeout 300 python3 -u -m pytest tests/test_rowchain_gpu.py tests/test_train_chains_gpu.py tests/test_timed_size_parity_gpu.py tests/test_modules_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
for v in 1 0; do
GD4D_MHA_PRESPLIT=$v timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout > $o/t_${v}_$rep.json 2> $o/t_${v}_$rep.err; echo "train dropout planes=$v $(ms $o/t_${v}_$rep.json)"
done
done
for v in 1 0; do
GD4D_MHA_PRESPLIT=$v timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline > $o/e_${v}.json 2> $o/e_${v}.err; echo "train eval planes=$v $(ms $o/e_${v}.json)"
done
