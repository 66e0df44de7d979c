#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run80; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("optimizer","")[:40])'; }
for rep in 1 2 3; do
timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout > $o/f_$rep.json 2> $o/f_$rep.err; echo "flat sgd  $(ms $o/f_$rep.json)"
timeout 200 python3 bench.py --mode train --steps 40 --warmup 3 --no-roofline --dropout --torch-sgd > $o/t_$rep.json 2> $o/t_$rep.err; echo "torch sgd $(ms $o/t_$rep.json)"
done
timeout 200 python3 bench.py --mode train --steps 10 --warmup 3 --no-roofline --criterion > $o/c.json 2> $o/c.err; echo "criterion $(ms $o/c.json)"
timeout 200 python3 bench.py --mode train --steps 10 --warmup 3 --no-roofline --no-graph > $o/e.json 2> $o/e.err; echo "eager $(ms $o/e.json)"
timeout 300 python3 -m pytest tests/test_configs_gpu.py tests/test_training_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
