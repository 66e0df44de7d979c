#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run64; mkdir -p $o
GD4D_CHECK_HANDOFF=1 timeout 900 python3 -m pytest tests/test_rowchain_gpu.py tests/test_modules_gpu.py tests/test_end_to_end_gpu.py tests/test_configs_gpu.py -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("value_batch1"))'; }
for rep in 1 2 3; do
for v in chaina chainb; do
GD4D_POS_ENCODER=$v python3 bench.py --inflight 1 --steps 200 --warmup 10 --no-roofline --no-cpu-baseline > $o/b_${v}_$rep.json 2> $o/b_${v}_$rep.err; echo "pos_encoder=$v $(ms $o/b_${v}_$rep.json)"
done
done
python3 tools/trace_step.py > $o/timeline.txt 2>&1; sed -n 18,34p $o/timeline.txt
