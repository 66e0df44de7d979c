#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run43; mkdir -p $o
timeout 900 python3 -m pytest tests/test_training_gpu.py tests/test_train_chains_gpu.py tests/test_timed_size_parity_gpu.py tests/test_configs_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -5
for rep in 1 2; do for v in bf16x3:x fp32:fp32; do name=${v%%:*}; val=${v##*:}
GD4D_WGRAD=$val python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/${name}_$rep.json 2> $o/${name}_$rep.err; echo "wgrad $name $(tail -1 $o/${name}_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
