#!/bin/bash
# what-if (timing only, wrong results): corners of the coarse levels served from LDS instead of the L1 path
cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/v; cp graph-detr4d_amd/libgd4d.so /tmp/v/base.so
for lv in 3 2 1; do
  touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="-DGD4D_WHATIF_LDS_LEVELS=$lv" 2>&1 | grep -i error
  cp graph-detr4d_amd/libgd4d.so /tmp/v/lds$lv.so
done
cp /tmp/v/base.so graph-detr4d_amd/libgd4d.so
for rep in 1 2; do for v in base lds3 lds2 lds1; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_sliced.py 2>/dev/null | tail -1 | sed "s/^/$v: /"; done; done
for v in base lds2 lds1; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_sliced.py --alias 2>/dev/null | tail -1 | sed "s/^/$v alias: /"; done
