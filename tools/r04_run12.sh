#!/bin/bash
# the whole GPU suite + smoke after the hand-off order change; requests in flight up to 6 (the hand-off must not dead-lock)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run12; mkdir -p $o
timeout 2400 python3 -m pytest tests -x -q -m gpu > $o/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $o/pytest_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for n in 1 2 4 6; do GD4D_CHECK_HANDOFF=1 timeout 600 python3 bench.py --inflight $n --no-stress --no-cpu-baseline --no-roofline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['channels_last_input']; print('inflight $n:', round(d['value'],1), 'samples/s;', round(d['ms_per_sample_batch1'],4), 'ms one at a time; channels-last', round(c['value'],1), round(c['ms_per_sample_batch1'],4))"; done | tee $o/inflight.txt
python3 tools/trace_step.py > $o/step_timeline_device.txt 2>&1; sed -n 28,46p $o/step_timeline_device.txt
