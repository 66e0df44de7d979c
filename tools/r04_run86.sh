#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for n in 2 3 4; do
echo "inflight $n: $(timeout 200 python3 bench.py --inflight $n --steps 120 --warmup 10 --no-roofline --no-cpu-baseline --no-nhwc-figure 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["ms_per_step"])')"
done
done
