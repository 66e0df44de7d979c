#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run66; mkdir -p $o
timeout 900 python3 -m pytest tests/test_rowchain_gpu.py tests/test_modules_gpu.py tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
python3 bench.py --inflight 1 --steps 200 --warmup 10 --no-roofline --no-cpu-baseline > $o/i_new_$rep.json 2>/dev/null; echo "infer new $(ms $o/i_new_$rep.json)"
GD4D_LIB_PATH=$PWD/build_ab/libgd4d_old.so python3 bench.py --inflight 1 --steps 200 --warmup 10 --no-roofline --no-cpu-baseline > $o/i_old_$rep.json 2>/dev/null; echo "infer old $(ms $o/i_old_$rep.json)"
python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/t_new_$rep.json 2>/dev/null; echo "train new $(ms $o/t_new_$rep.json)"
GD4D_LIB_PATH=$PWD/build_ab/libgd4d_old.so python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/t_old_$rep.json 2>/dev/null; echo "train old $(ms $o/t_old_$rep.json)"
done
