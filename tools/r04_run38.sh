#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run38; mkdir -p $o
ulimit -c 0
for i in 1 2; do timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/pytest_$i.log 2>&1; echo "all $i rc=$? $(tail -1 $o/pytest_$i.log)"; done
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
python3 bench.py > $o/bench.json 2> $o/bench.err; tail -1 $o/bench.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("default", d["value"], d.get("value_batch1"), d["roofline"]["frac"], d["roofline"]["traffic"])'
