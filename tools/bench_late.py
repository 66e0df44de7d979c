#!/usr/bin/env python
"""Full-size timing of the aggregate-then-project kernels (dev tool; bench.py is the contract):
gd4d_pyramid_channels_last_fwd, gd4d_cross_attn_agg_fwd, gd4d_value_proj_heads_fwd."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402


def timed(fn, iters, launches_per_call=1):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / launches_per_call * 1e3     # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--all-visible', action='store_true')
    ap.add_argument('--no-order', action='store_true')
    ap.add_argument('--levels', default='r50')
    ap.add_argument('--split', action='store_true')
    a = ap.parse_args()
    dev = 'cuda'
    n, q = 6 * a.frames, a.queries
    levels = synthetic.R50_LEVELS if a.levels == 'r50' else synthetic.VOV_LEVELS
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in levels]
    rig = synthetic.camera_rig(a.frames)
    if a.all_visible:
        rig[:] = rig[0]
    l2i = torch.from_numpy(rig).unsqueeze(0).to(dev)
    ref = torch.rand(1, q, 3, generator=g)
    if a.all_visible:
        ref[..., 0] = 0.6 + 0.3 * ref[..., 0]
        ref[..., 1] = 0.45 + 0.1 * ref[..., 1]
        ref[..., 2] = 0.6 + 0.1 * ref[..., 2]
    ref = ref.to(dev)
    offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
    attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(1, q, n, generator=g).to(dev)
    w = (torch.randn(256, 256, generator=g) * 0.06).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    order = None if a.no_order else ops.query_order_fwd(ref, synthetic.PC_RANGE)
    cl, shapes = ops.pyramid_channels_last_fwd(feats)
    agg, wsum, mask = ops.cross_attn_agg_fwd(cl, shapes, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8,
                                             want_mask=True, query_order=order)
    vis = int(mask.sum().item())
    nbytes = cl.numel() * 4
    t_cl = timed(lambda: ops.pyramid_channels_last_fwd(feats, out=cl), a.iters)
    # the six layers of a decoder read the SAME channels-last tensor (757 MB >> 256 MB Infinity Cache)
    t_agg = timed(lambda: [ops.cross_attn_agg_fwd(cl, shapes, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8,
                                                  query_order=order) for _ in range(6)], a.iters, 6)
    t_hp = timed(lambda: [ops.value_proj_heads_fwd(agg, wsum, w, bias) for _ in range(6)], a.iters, 6)
    tuples = vis * len(levels)
    alg_raw = tuples * 4 * 1024
    print(f'N={n} Q={q} visible (cam,q,h,p)={vis} ({vis / mask.numel():.3f})  tuples x levels={tuples}')
    print(f'channels-last copy: {t_cl:.1f} us  ({2 * nbytes / t_cl / 1e6:.2f} TB/s read+write of {nbytes / 1e6:.0f} MB)')
    print(f'aggregate (raw gather): {t_agg:.1f} us per layer;  corner bytes {alg_raw / 1e6:.0f} MB -> {alg_raw / t_agg / 1e6:.2f} TB/s at the L2; '
          f'capped at the tensor {min(alg_raw, nbytes) / 1e6:.0f} MB -> {min(alg_raw, nbytes) / t_agg / 1e6:.2f} TB/s')
    print(f'value_proj of the aggregates: {t_hp:.1f} us per layer')
    # channel-sliced form: slice-planar copy, plan, gather
    sp, _ = ops.pyramid_slice_planar_fwd(feats)
    pyr = ops.PyramidView.slice_planar(sp, shapes)
    t_sp = timed(lambda: ops.pyramid_slice_planar_fwd(feats, out=sp), a.iters)
    t_sp_p = timed(lambda: ops.pyramid_slice_planar_fwd(feats, out=sp, max_cus=224), a.iters)
    plan = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order)
    t_plan = timed(lambda: [ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, plan=plan, query_order=order)
                            for _ in range(6)], a.iters, 6)
    sa, sw = ops.cross_attn_agg_sliced_fwd(plan), plan.wsum
    print(f'sliced vs rows: max |agg diff| {(sa - agg).abs().max().item():.2e}  max |wsum diff| {(sw - wsum).abs().max().item():.2e}')
    t_sl = timed(lambda: [ops.cross_attn_agg_sliced_fwd(plan, agg=sa) for _ in range(6)], a.iters, 6)
    print(f'slice-planar copy: {t_sp:.1f} us  (persistent on 224 CUs: {t_sp_p:.1f} us)')
    print(f'plan: {t_plan:.1f} us per layer')
    print(f'sliced aggregate: {t_sl:.1f} us per layer  ({alg_raw / t_sl / 1e6:.2f} TB/s at the L2)')
    pm = ops.PyramidView.pixel_major(cl, shapes)
    plan_pm = ops.cross_attn_plan_fwd(pm, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order)
    t_pm = timed(lambda: [ops.cross_attn_agg_sliced_fwd(plan_pm, agg=sa) for _ in range(6)], a.iters, 6)
    print(f'sliced aggregate on the pixel-major copy (= caller-owned NHWC levels): {t_pm:.1f} us per layer')
    if a.split:
        for parts in ([(0, 4), (4, 4)], [(i, 1) for i in range(8)]):
            t = timed(lambda: [[ops.cross_attn_agg_sliced_fwd(plan, slices=sl, agg=sa) for sl in parts]
                               for _ in range(6)], a.iters, 6)
            print(f'sliced aggregate in {len(parts)} launches: {t:.1f} us per layer')
    print(f'six layers: {t_cl + 6 * (t_agg + t_hp):.0f} us')


if __name__ == '__main__':
    main()
