#!/bin/bash
# plan workgroups mapped to their XCD's sector: A/B
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run6; mkdir -p $o
timeout 600 python3 -m pytest tests/test_cross_attn_sliced_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
b1() { python3 bench.py --inflight $2 --no-stress --no-cpu-baseline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 inflight $2:', round(d['ms_per_sample_batch1'],4), 'ms per sample one at a time,', round(d['value'],1), 'samples/s; gather', round(r['us_per_launch'],1), 'us; plan', round(d['kernels']['cross_attn_plan']['us_per_launch'],1))"; }
b1 xcd 1; b1 xcd 1; python3 tools/bench_sliced.py | tail -1
touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
make -s -C graph-detr4d_amd/csrc EXTRA="-DGD4D_PLAN_XCD=0" 2>&1 | grep -i error
b1 r3map 1; b1 r3map 1; python3 tools/bench_sliced.py | tail -1
touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
make -s -C graph-detr4d_amd/csrc 2>&1 | grep -i error
b1 xcd 1
