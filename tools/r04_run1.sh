#!/bin/bash
# round 4, run 1: items-form gather - parity, A/B timing against the pairs form, PMC traffic, a bench line
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run1; mkdir -p $o
python3 -m pytest tests/test_cross_attn_sliced_gpu.py tests/test_abi.py -x -q -m gpu > $o/pytest_sliced.log 2>&1; echo "pytest sliced rc=$?"; tail -3 $o/pytest_sliced.log
python3 -m pytest tests/test_configs_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "config4 or queued or accumulated" > $o/pytest_train.log 2>&1; echo "pytest train rc=$?"; tail -3 $o/pytest_train.log
for plan in pairs items pairs items; do python3 tools/bench_sliced.py --plan $plan 2>&1 | tail -1; done | tee $o/ab.txt
python3 tools/bench_sliced.py --plan items --alias 2>&1 | tail -1 | tee -a $o/ab.txt
python3 tools/bench_sliced.py --plan pairs --alias 2>&1 | tail -1 | tee -a $o/ab.txt
python3 tools/bench_sliced.py --plan items --layout pixel 2>&1 | tail -1 | tee -a $o/ab.txt
python3 tools/bench_sliced.py --plan items --dtype bf16 2>&1 | tail -1 | tee -a $o/ab.txt
bash tools/prof_pmc_sliced.sh r04_run1 2>&1 | tail -20
python3 bench.py --no-stress > $o/bench.json 2> $o/bench.err; tail -c 600 $o/bench.json
python3 bench.py --inflight 1 --no-stress --no-cpu-baseline > $o/bench1.json 2> $o/bench1.err; python3 - <<PY
import json
l=json.loads(open('$o/bench1.json').read().strip().splitlines()[-1])
print('batch1', l['value_batch1'], l['ms_per_sample_batch1'], 'roofline', {k:l['roofline'][k] for k in ('us_per_launch','frac','traffic')}, 'plan', l['kernels'].get('cross_attn_plan'))
PY
