#!/usr/bin/env python
"""Dev: would the forward gather and the record count overlap if they ran at the same time?  (two streams, eager)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402

dev = 'cuda'
n, q = 24, 900
levels = synthetic.R50_LEVELS
g = torch.Generator().manual_seed(0)
feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in levels]
l2i = torch.from_numpy(synthetic.camera_rig(4)).unsqueeze(0).to(dev)
ref = torch.rand(1, q, 3, generator=g).to(dev)
offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
cam = torch.randn(1, q, n, generator=g).to(dev)
order = ops.query_order_fwd(ref, synthetic.PC_RANGE)
sp, shapes = ops.pyramid_slice_planar_fwd(feats)
pyr = ops.PyramidView.slice_planar(sp, shapes)
del feats
plan = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order)
sa = ops.cross_attn_agg_sliced_fwd(plan)
sink = ops.PyramidGrad(pyr, 64, 1, q, 8, chunk_walk=False)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode, reps=20):
    torch.cuda.synchronize()
    sink.count.zero_()
    sink.plans = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for i in range(reps):
        if mode in ('gather', 'both', 'serial'):
            with torch.cuda.stream(s1):
                ops.cross_attn_agg_sliced_fwd(plan, agg=sa)
        if mode in ('count', 'both'):
            with torch.cuda.stream(s2):
                sink.add_layer(i, plan)
        if mode == 'serial':
            with torch.cuda.stream(s1):
                sink.add_layer(i, plan)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for mode in ('gather', 'count', 'serial', 'both', 'gather', 'count', 'serial', 'both'):
    print(f'{mode:7s}: {run(mode):7.1f} us per layer')
