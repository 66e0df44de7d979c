#!/bin/bash
# attention core (bf16x3): look-ahead and waves-per-workgroup variants
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run10; mkdir -p $o
timeout 600 python3 -m pytest tests/test_dense_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
b1() { python3 bench.py --inflight 1 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_sample_batch1'],4), 'ms per sample')"; python3 tools/trace_step.py 2>/dev/null | grep "end   mha_core" | sed -n 3,4p; }
b1 lookahead-8waves; b1 lookahead-8waves
for v in "-DMHA_LOOKAHEAD=0" "-DMHA_WAVES_N=16" "-DMHA_WAVES_N=16 -DMHA_LOOKAHEAD=0" "-DMHA_WAVES_N=4"; do
  touch graph-detr4d_amd/csrc/gd4d_self_attn.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  b1 "$v"; b1 "$v"
done
touch graph-detr4d_amd/csrc/gd4d_self_attn.hip; make -s -C graph-detr4d_amd/csrc 2>&1 | grep -i error
