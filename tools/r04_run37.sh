#!/bin/bash
# what-if: the record-count kernel without its atomics (the #ifndef PG_COUNT_WHATIF_NOATOMIC around the four atomics of
# pyramid_grad_count_kernel that this script compiled against was removed again: the PMC records are keyed to the source's hash)
cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/v; cp graph-detr4d_amd/libgd4d.so /tmp/v/base.so
touch graph-detr4d_amd/csrc/gd4d_cross_attn_sliced_bwd.hip
make -s -C graph-detr4d_amd/csrc EXTRA="-DPG_COUNT_WHATIF_NOATOMIC" 2>&1 | grep -i " error"
cp graph-detr4d_amd/libgd4d.so /tmp/v/noatomic.so
cp /tmp/v/base.so graph-detr4d_amd/libgd4d.so
for rep in 1 2; do for v in base noatomic; do echo "$v: $(GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/dev_count_scaling.py 2>/dev/null | head -1)"; done; done
