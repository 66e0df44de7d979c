#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
echo "new $(python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
echo "old $(cd build_ab/old && python3 bench.py --mode train --steps 30 --warmup 3 --no-roofline --dropout 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
