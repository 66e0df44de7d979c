#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run81; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("optimizer","")[:30])'; }
timeout 200 python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout > $o/f.json 2> $o/f.err; echo "flat sgd  $(ms $o/f.json)"
timeout 200 python3 bench.py --mode train --steps 10 --warmup 3 --no-roofline --criterion > $o/c.json 2> $o/c.err; echo "criterion $(ms $o/c.json)"; tail -2 $o/c.err | cut -c1-200
timeout 200 python3 bench.py --mode distill --steps 5 --warmup 2 --no-roofline > $o/d.json 2> $o/d.err; echo "distill $(ms $o/d.json)"; tail -1 $o/d.err | cut -c1-200
timeout 200 python3 bench.py --mode train --steps 5 --warmup 2 --no-roofline --levels vov > $o/v.json 2> $o/v.err; echo "vov $(ms $o/v.json)"
GD4D_TRAIN_CHAINS=0 timeout 200 python3 bench.py --mode train --steps 10 --warmup 3 --no-roofline --dropout > $o/g.json 2> $o/g.err; echo "generic $(ms $o/g.json)"
ulimit -c 0
timeout 420 python3 -u -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/all.log 2>&1; echo "all rc=$? $(tail -1 $o/all.log)"; grep -n "^E " $o/all.log | head -6
