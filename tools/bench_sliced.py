#!/usr/bin/env python
"""Full-size timing of the channel-sliced gather alone (dev tool; bench.py is the contract).
  --alias      all camera rows alias row 0 (cam_stride = 0): a phase's footprint is one camera's slice (3.9 MB) - the
               kernel's own floor with the L2 misses taken away
  --layout     planar | pixel   the slice-planar copy or the pixel-major copy (= caller-owned channels-last levels)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402
from bench_late import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--alias', action='store_true')
    ap.add_argument('--no-order', action='store_true')
    ap.add_argument('--layout', default='planar')
    ap.add_argument('--levels', default='r50')
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--plan', default='items', help='items (the inference step) | pairs (what the training kernels read)')
    ap.add_argument('--coarse', action='store_true', help='also time gd4d_cross_attn_agg_items_coarse_fwd and the projection of levels 2-3')
    ap.add_argument('--footprint', action='store_true',
                    help='count the distinct pixels the visible samples touch per level: the bytes a gather has to bring in at least once')
    a = ap.parse_args()
    dev = 'cuda'
    n, q = 6 * a.frames, a.queries
    levels = synthetic.R50_LEVELS if a.levels == 'r50' else synthetic.VOV_LEVELS
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in levels]
    l2i = torch.from_numpy(synthetic.camera_rig(a.frames)).unsqueeze(0).to(dev)
    ref = torch.rand(1, q, 3, generator=g).to(dev)
    offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
    attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(1, q, n, generator=g).to(dev)
    order = None if a.no_order else ops.query_order_fwd(ref, synthetic.PC_RANGE)
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    if a.layout == 'planar':
        sp, shapes = ops.pyramid_slice_planar_fwd(feats, out_dtype=dt)
        pyr = ops.PyramidView.slice_planar(sp, shapes)
    else:
        cl, shapes = ops.pyramid_channels_last_fwd(feats, out_dtype=dt)
        pyr = ops.PyramidView.pixel_major(cl, shapes)
    coarse = None
    if a.coarse:
        wv, bv = (torch.randn(256, 256, generator=g) * 0.06).to(dev), torch.randn(256, generator=g).to(dev)
        cf = [f.contiguous() for f in feats[2:]]
        proj = ops.value_proj_fwd(cf, wv, bv)
        coarse = ops.CoarseValues(proj, [tuple(f.shape[-2:]) for f in cf])
        t_proj = timed(lambda: [ops.value_proj_fwd(cf, wv, bv, out=proj) for _ in range(6)], a.iters, 6)
        print(f'value_proj over levels 2-3 ({proj.shape[0] * proj.shape[1]} rows, weight image launch included): {t_proj:.1f} us')
    del feats
    if a.alias:
        pyr.cam_stride = [0] * len(pyr.cam_stride)
    items = a.plan == 'items'
    plan, mask = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8,
                                         query_order=order, want_mask=True, items=items)
    vis = int(mask.sum().item())
    if a.footprint:
        # mmcv's MSDA / grid_sample(align_corners=False): pixel x = u W - 0.5; the four corners around it, those inside the map
        _, m2, uv = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order,
                                            want_mask=True, want_uv=True, items=items)
        keep = m2.bool().reshape(-1)                                   # (B, N, Q, Hh, P)
        camid = torch.arange(n, device=dev).view(1, n, 1, 1, 1).expand_as(m2).reshape(-1)[keep]
        headid = torch.arange(8, device=dev).view(1, 1, 1, 8, 1).expand_as(m2).reshape(-1)[keep]
        u, v = uv[..., 0].reshape(-1)[keep], uv[..., 1].reshape(-1)[keep]
        total_raw = total_proj = 0
        for lvl, (h, w) in enumerate(levels):
            x0, y0 = torch.floor(u * w - 0.5).long(), torch.floor(v * h - 0.5).long()
            pix, pix_h = [], []
            for dy in (0, 1):
                for dx in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
                    key = (camid[ok] * h + yy[ok]) * w + xx[ok]
                    pix.append(key)
                    pix_h.append(key * 8 + headid[ok])
            upix, uph = torch.unique(torch.cat(pix)).numel(), torch.unique(torch.cat(pix_h)).numel()
            total_raw += upix * 1024
            total_proj += uph * 128
            print(f'level {lvl} ({h} x {w}): {upix} of {n * h * w} pixels touched = {upix * 1024 / 1e6:.1f} MB of raw 1-KB rows; '
                  f'{uph} (pixel, head) pairs = {uph * 128 / 1e6:.1f} MB of projected 128-byte rows')
        print(f'at least once: all levels raw {total_raw / 1e6:.1f} MB')
    corner_bytes = vis * len(levels) * 4 * 256 * (2 if a.dtype == 'bf16' else 4)
    sa, sw = ops.cross_attn_agg_sliced_fwd(plan), plan.wsum
    t_plan = timed(lambda: [ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8,
                                                    plan=plan, query_order=order, items=items) for _ in range(6)], a.iters, 6)
    t = timed(lambda: [ops.cross_attn_agg_sliced_fwd(plan, agg=sa) for _ in range(6)], a.iters, 6)
    if coarse is not None:
        ca, cp = ops.cross_attn_agg_coarse_fwd(plan, coarse)
        tc = timed(lambda: [ops.cross_attn_agg_coarse_fwd(plan, coarse, agg=ca, pagg=cp) for _ in range(6)], a.iters, 6)
        ops.cross_attn_agg_sliced_fwd(plan, agg=sa)
        full = ops.value_proj_heads_fwd(sa, plan.wsum, wv, bv)
        ops.cross_attn_agg_coarse_fwd(plan, coarse, agg=ca, pagg=cp)
        mixed = ops.value_proj_heads_fwd(ca, plan.wsum, wv, bv) + cp
        print(f'coarse-projected gather {tc:.1f} us per launch; max |difference| to the raw gather {float((mixed - full).abs().max()):.2e}')
    print(f'plan {a.plan} layout {a.layout} alias {a.alias} {a.dtype}: plan {t_plan:.1f} us, '
          f'sliced gather {t:.1f} us per launch ({corner_bytes / t / 1e6:.2f} TB/s of corner bytes)')


if __name__ == '__main__':
    main()
