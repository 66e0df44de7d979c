cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/v
for a in 0 1; do for t in 0 1; do
  touch graph-detr4d_amd/csrc/gd4d_mlp2.hip
  make -s -C graph-detr4d_amd/csrc EXTRA="-DML_DMA_IN_A=$a -DML_TWO_ACC=$t" 2>&1 | grep -i error
  cp graph-detr4d_amd/libgd4d.so /tmp/v/m$a$t.so
done; done
for rep in 1 2; do for v in m00 m01 m10 m11; do GD4D_LIB_PATH=/tmp/v/$v.so python3 tools/bench_mlp2.py 2>/dev/null | head -1 | sed "s/^/$v: /"; done; done
