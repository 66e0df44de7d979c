#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run11; mkdir -p $o
timeout 900 python3 -m pytest tests/test_modules_gpu.py tests/test_full_size_gpu.py tests/test_end_to_end_gpu.py tests/test_rowchain_gpu.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.log
b1() { python3 bench.py --inflight $2 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 inflight $2:', round(d['ms_per_sample_batch1'],4), 'ms per sample one at a time,', round(d['value'],1), 'samples/s')"; }
b1 early 1; GD4D_REG_EARLY=0 b1 late 1; b1 early 1; GD4D_REG_EARLY=0 b1 late 1; b1 early 2; GD4D_REG_EARLY=0 b1 late 2
python3 tools/trace_step.py > $o/step_timeline_device.txt 2>&1; sed -n 28,40p $o/step_timeline_device.txt
