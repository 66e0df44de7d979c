#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run82; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2 3; do
timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout > $o/n_$rep.json 2> $o/n_$rep.err; echo "new $(ms $o/n_$rep.json)"
done
timeout 300 python3 -u -m pytest tests/test_train_chains_gpu.py tests/test_training_gpu.py tests/test_modules_gpu.py tests/test_end_to_end_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
