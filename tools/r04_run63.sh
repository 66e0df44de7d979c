#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run63; mkdir -p $o
GD4D_CHECK_HANDOFF=1 timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"; grep -n "^E " $o/tests.log | head -8
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("value"))'; }
for v in 1 0; do
GD4D_MHA_PRESPLIT=$v python3 bench.py --queries 2700 --inflight 1 --steps 50 --warmup 5 --no-roofline --no-cpu-baseline > $o/h_${v}.json 2> $o/h_${v}.err; echo "hdetr presplit=$v $(ms $o/h_${v}.json)"
GD4D_MHA_PRESPLIT=$v python3 bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline > $o/d_${v}.json 2> $o/d_${v}.err; echo "default (2 in flight) presplit=$v $(ms $o/d_${v}.json)"
done
