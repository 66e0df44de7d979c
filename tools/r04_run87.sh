#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for st in 0 3 6 9 12 16; do
echo "stagger $st: $(GD4D_BENCH_STAGGER=$st timeout 200 python3 bench.py --inflight 2 --steps 120 --warmup 10 --no-roofline --no-cpu-baseline --no-nhwc-figure 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["ms_per_step"])')"
done
