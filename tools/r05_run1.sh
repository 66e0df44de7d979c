#!/bin/bash
# round 5, run 1 (what-if, timing only, wrong results): the items gather with the loads of levels >= K skipped - the floor of a gather
# that keeps levels 0..K-1 on today's walk and hands the coarse levels to a de-duplicating walk
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_run1; mkdir -p $o
mkdir -p /tmp/v; cp graph-detr4d_amd/libgd4d.so /tmp/v/base.so
for lv in 3 2 1; do
  touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="-DGD4D_WHATIF_A_LEVELS=$lv" 2>&1 | grep -i error
  cp graph-detr4d_amd/libgd4d.so /tmp/v/a$lv.so
done
cp /tmp/v/base.so graph-detr4d_amd/libgd4d.so
for rep in 1 2; do for v in base a3 a2 a1; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_sliced.py 2>/dev/null | tail -1 | sed "s/^/$v: /"; done; done | tee $o/ab.txt
for v in base a2; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_sliced.py --alias 2>/dev/null | tail -1 | sed "s/^/$v alias: /"; done | tee -a $o/ab.txt
for v in base a2; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_sliced.py --layout pixel 2>/dev/null | tail -1 | sed "s/^/$v pixel: /"; done | tee -a $o/ab.txt
