"""Premise check for 'early projection of the coarse levels' (VERDICT r3 top_next, option c): what does value_proj over levels 2-3
only (43 800 pixels at 24 cameras) cost - one layer, and five layers in one launch?  Alone and beside a gather on another stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_detr4d_amd import ops, synthetic


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev = 'cuda'
n = 24
g = torch.Generator().manual_seed(0)
feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in synthetic.R50_LEVELS]
ws = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(6)]
bs = [torch.randn(256, generator=g).to(dev) for _ in range(6)]
coarse = feats[2:]
for levels, name in ((coarse, 'levels 2-3'), (feats[3:], 'level 3'), (feats[1:], 'levels 1-3')):
    t1 = timed(lambda: ops.value_proj_fwd(levels, ws[0], bs[0]))
    t5 = timed(lambda: ops.value_proj_multi_fwd(levels, ws[1:], bs[1:]))
    t6 = timed(lambda: ops.value_proj_multi_fwd(levels, ws, bs))
    print(f'{name}: one layer {t1:.1f} us, five layers in one launch {t5:.1f} us, six {t6:.1f} us', flush=True)
# beside a gather (second stream)
sp, shapes = ops.pyramid_slice_planar_fwd(feats)
pyr = ops.PyramidView.slice_planar(sp, shapes)
l2i = torch.from_numpy(synthetic.camera_rig(4)).unsqueeze(0).to(dev)
q = 900
ref = torch.rand(1, q, 3, generator=g).to(dev)
off = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
att = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
cam = torch.randn(1, q, n, generator=g).to(dev)
order = ops.query_order_fwd(ref, synthetic.PC_RANGE)
plan = ops.cross_attn_plan_fwd(pyr, ref, off, att, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order, items=True)
agg = ops.cross_attn_agg_sliced_fwd(plan)
side = torch.cuda.Stream()
tg = timed(lambda: ops.cross_attn_agg_sliced_fwd(plan, agg=agg))


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.value_proj_multi_fwd(coarse, ws[1:], bs[1:])
    ops.cross_attn_agg_sliced_fwd(plan, agg=agg)
    torch.cuda.current_stream().wait_stream(side)
tb = timed(both)
print(f'gather alone {tg:.1f} us; gather + five-layer coarse projection on a second stream {tb:.1f} us', flush=True)
