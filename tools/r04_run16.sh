#!/bin/bash
# LDS stage for the coarse levels' corners: parity, then timing stand-alone and in the step (GD4D_STAGE)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run16; mkdir -p $o
timeout 900 python3 -m pytest tests/test_cross_attn_sliced_gpu.py tests/test_abi.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $o/pytest.log
for rep in 1 2; do for st in 0 3 2; do python3 tools/bench_sliced.py --stage $st 2>/dev/null | tail -1; done; done | tee $o/ab.txt
b1() { GD4D_STAGE=$1 python3 bench.py --inflight 1 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stage $1:', round(d['ms_per_sample_batch1'],4), 'ms per sample')"; }
for rep in 1 2; do b1 0; b1 3; b1 2; done | tee -a $o/ab.txt
