"""Does gd4d_pyramid_grad_count scale with the number of (query, head) waves of ONE launch, or is it bound by the latency of a
wave's chain?  Times the count (and fill) kernels for Q = 900 and for one plan of Q = 5400 (= six layers' worth in one launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_detr4d_amd import ops, synthetic


def timed(fn, iters=20):
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
frames = 4
n = 6 * frames
g = torch.Generator().manual_seed(0)
feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in synthetic.R50_LEVELS]
sp, shapes = ops.pyramid_slice_planar_fwd(feats)
pyr = ops.PyramidView.slice_planar(sp, shapes)
l2i = torch.from_numpy(synthetic.camera_rig(frames)).unsqueeze(0).to(dev)
for q in (900, 1800, 3600):
    ref = torch.rand(1, q, 3, generator=g).to(dev)
    off = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
    att = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(1, q, n, generator=g).to(dev)
    order = ops.query_order_fwd(ref, synthetic.PC_RANGE)
    plan = ops.cross_attn_plan_fwd(pyr, ref, off, att, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order)
    sink = ops.PyramidGrad(pyr, 1, 1, q, 8, chunk_walk=False)

    def count():
        sink.plans = []
        sink.add_layer(0, plan)
    t = timed(count)
    print(f'Q = {q}: count {t:.1f} us ({t / (q / 900):.1f} us per 900 queries)', flush=True)
