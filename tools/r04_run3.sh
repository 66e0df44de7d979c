#!/bin/bash
# round 4, run 3: SIGNAL / WAIT hand-off + position_encoder beside chain B: parity, then A/B against round 3's schedule
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run3; mkdir -p $o
timeout 600 python3 -m pytest tests/test_rowchain_gpu.py -x -q -m gpu > $o/pytest_chain.log 2>&1; echo "pytest chain rc=$?"; tail -5 $o/pytest_chain.log
timeout 900 python3 -m pytest tests/test_modules_gpu.py tests/test_full_size_gpu.py tests/test_end_to_end_gpu.py -x -q -m gpu > $o/pytest_mod.log 2>&1; echo "pytest modules rc=$?"; tail -5 $o/pytest_mod.log
b1() { python3 bench.py --inflight $2 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 inflight $2:', round(d['ms_per_sample_batch1'],4), 'ms per sample one at a time,', round(d['value'],1), 'samples/s')"; }
b1 chainb 1; GD4D_POS_ENCODER=dual b1 dual 1; b1 chainb 1; GD4D_POS_ENCODER=dual b1 dual 1
b1 chainb 2; GD4D_POS_ENCODER=dual b1 dual 2
python3 tools/trace_step.py > $o/step_timeline_device.txt 2>&1; tail -60 $o/step_timeline_device.txt
