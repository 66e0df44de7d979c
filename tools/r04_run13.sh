#!/bin/bash
# gather: slice stagger across XCDs, non-temporal store of the aggregates (dev A/B; rebuilds gd4d_cross_attn_sliced.o on the box)
cd "$GRAFT_REPO_ROOT"
b1() { python3 tools/bench_sliced.py 2>/dev/null | tail -1 | sed "s/^/$1: /"; python3 bench.py --inflight 1 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_sample_batch1'],4), 'ms per sample')"; }
b1 base; b1 base
for v in "-DGD4D_SLICE_STAGGER" "-DGD4D_AGG_NT_STORE" "-DGD4D_SLICE_STAGGER -DGD4D_AGG_NT_STORE"; do
  touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  b1 "$v"; b1 "$v"
done
touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip; make -s -C graph-detr4d_amd/csrc 2>&1 | grep -i error
