#!/bin/bash
# round 4, run 4: the whole GPU suite + smoke, inflight sweep
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run4; mkdir -p $o
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=15 > $o/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -25 $o/pytest_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for n in 1 2 3 4; do python3 bench.py --inflight $n --no-stress --no-cpu-baseline --no-roofline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['channels_last_input']; print('inflight $n:', round(d['value'],1), 'samples/s;', round(d['ms_per_sample_batch1'],4), 'ms one at a time; channels-last', round(c['value'],1), round(c['ms_per_sample_batch1'],4))"; done | tee $o/inflight.txt
