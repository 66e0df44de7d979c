#!/bin/bash
# gather variants, interleaved runs from prebuilt libraries (GD4D_LIB_PATH) so that run order does not decide
cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/v
cp graph-detr4d_amd/libgd4d.so /tmp/v/base.so
i=0
for v in "-DGD4D_SLICE_STAGGER" "-DGD4D_AGG_NT_STORE" "-DGD4D_SLICE_STAGGER -DGD4D_AGG_NT_STORE"; do
  i=$((i+1))
  touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  cp graph-detr4d_amd/libgd4d.so /tmp/v/v$i.so
done
cp /tmp/v/base.so graph-detr4d_amd/libgd4d.so
b1() { GD4D_LIB_PATH=/tmp/v/$1.so python3 bench.py --inflight $2 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 80 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 inflight $2:', round(d['ms_per_sample_batch1'],4), 'ms per sample;', round(d['value'],1))"; }
for rep in 1 2 3; do for v in base v3 v1 v2; do b1 $v 1; done; done
b1 base 2; b1 v3 2; b1 base 2; b1 v3 2
GD4D_LIB_PATH=/tmp/v/v3.so python3 -m pytest tests/test_cross_attn_sliced_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -2
