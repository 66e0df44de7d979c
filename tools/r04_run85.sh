#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run85; mkdir -p $o
ulimit -c 0
for i in 1 2 3; do
timeout 300 python3 -u -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/all_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $o/all_$i.log)"
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python3 bench.py 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default:', round(d['value'],1), d.get('value_batch1'), d['roofline']['frac'], d['roofline']['traffic'])"
