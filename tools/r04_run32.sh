#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run32; mkdir -p $o
for rep in 1 2 3; do
for v in split:graph-detr4d_amd/libgd4d.so nosplit:build_ab/libgd4d_nosplit.so; do
name=${v%%:*}; lib=${v##*:}
GD4D_LIB_PATH=$GRAFT_REPO_ROOT/$lib python3 bench.py --inflight 1 --no-stress --no-roofline --no-cpu-baseline --steps 60 --warmup 5 > $o/${name}_$rep.json 2> $o/${name}_$rep.err
echo "$name $(tail -1 $o/${name}_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["value"],1))')"
done; done
timeout 900 python3 -m pytest tests/test_rowchain_gpu.py tests/test_train_chains_gpu.py tests/test_modules_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -1 $o/bench_default.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("default", d["value"], d.get("value_batch1"), d["roofline"]["frac"], d["roofline"]["traffic"])'
