#!/usr/bin/env python
"""Quick full-size timing of gd4d_cross_attn_fwd (dev tool; bench.py is the contract)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--iters', type=int, default=60)
    ap.add_argument('--all-visible', action='store_true')
    ap.add_argument('--head-major', action='store_true')
    ap.add_argument('--order', action='store_true', help='pass gd4d_query_order_fwd order to the kernel')
    ap.add_argument('--order-azimuth', action='store_true', help='pass a host-side azimuth argsort as the order')
    ap.add_argument('--sort-queries', action='store_true', help='order the queries by azimuth (locality probe)')
    ap.add_argument('--rotate', type=int, default=6,
                    help='distinct value tensors to rotate through (cache-cold like the real decoder; 1 = warm)')
    a = ap.parse_args()
    dev = 'cuda'
    n, q = 6 * a.frames, a.queries
    levels = synthetic.R50_LEVELS
    s = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(0)
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    val = torch.randn(n, s, 8, 32, generator=g).to(dev, dt)
    vals = [val] + [torch.randn_like(val) for _ in range(a.rotate - 1)]
    if a.head_major:
        vals = [v.permute(0, 2, 1, 3).contiguous() for v in vals]
        val = vals[0]
    hm = dict(head_major=a.head_major)
    rig = synthetic.camera_rig(a.frames)
    if a.all_visible:
        rig[:] = rig[0]
    l2i = torch.from_numpy(rig).unsqueeze(0).to(dev)
    ref = torch.rand(1, q, 3, generator=g)
    if a.all_visible:                      # park every query in front of camera 0
        ref[..., 0] = 0.6 + 0.3 * ref[..., 0]
        ref[..., 1] = 0.45 + 0.1 * ref[..., 1]
        ref[..., 2] = 0.6 + 0.1 * ref[..., 2]
    if a.sort_queries:
        ang = torch.atan2(ref[0, :, 1] - 0.5, ref[0, :, 0] - 0.5)
        ref = ref[:, torch.argsort(ang)].contiguous()
    ref = ref.to(dev)
    offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
    attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(1, q, n, generator=g).to(dev)
    if a.order:
        hm['query_order'] = ops.query_order_fwd(ref, synthetic.PC_RANGE)
    if a.order_azimuth:
        ang = torch.atan2(ref[0, :, 1] - 0.5, ref[0, :, 0] - 0.5)
        hm['query_order'] = torch.argsort(ang).int()
    out, mask = ops.cross_attn_fwd(val, levels, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600,
                                   want_mask=True, **hm)
    vis = int(mask.sum().item())
    es = val.element_size()
    alg = vis * 4 * 4 * 32 * es + q * (3 + 96 + 128 + n) * 4 + n * 64 + q * 256 * 4
    alg = min(alg, val.numel() * es + q * 256 * 4)
    def launches():
        for v in vals:
            ops.cross_attn_fwd(v, levels, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, out=out, **hm)
    for _ in range(3):
        launches()
    # one hipGraph of len(vals) launches (eager calls of a 40-us kernel are close to host-bound)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launches()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        launches()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = max(1, a.iters // len(vals))
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    a.iters = reps * len(vals)
    ms = e0.elapsed_time(e1) / a.iters
    print(f'order={int(a.order) + 2 * int(a.order_azimuth)} sorted={int(a.sort_queries)} head_major={int(a.head_major)} rotate={a.rotate} N={n} Q={q} {a.dtype} visible(h,p) tuples={vis} ({vis / mask.numel():.3f}) '
          f'alg_bytes={alg / 1e6:.1f} MB  {ms * 1e3:.1f} us  {alg / ms / 1e9:.2f} TB/s  '
          f'frac_of_8TB/s={alg / ms / 1e9 / 8:.3f}')


if __name__ == '__main__':
    main()
