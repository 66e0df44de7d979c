#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run61; mkdir -p $o
for n in 2 4; do
rocprofv3 --kernel-trace -f csv -d $o/t$n -o t -- python3 bench.py --inflight $n --steps 60 --warmup 5 --no-roofline --no-cpu-baseline --no-nhwc-figure > $o/b$n.json 2> $o/b$n.err
t=$(find $o/t$n -name '*kernel_trace.csv' | head -1)
echo "inflight $n: $(tail -1 $o/b$n.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
python3 tools/dev_overlap_stats.py $t
find $o/t$n -name '*.csv' -delete
done
