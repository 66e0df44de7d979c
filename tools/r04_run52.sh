#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run52; mkdir -p $o
timeout 900 python3 -m pytest tests/test_train_chains_gpu.py tests/test_training_gpu.py tests/test_timed_size_parity_gpu.py tests/test_configs_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -5
for rep in 1 2 3; do
python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/new_$rep.json 2> $o/new_$rep.err; echo "chains $(tail -1 $o/new_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
