#!/usr/bin/env python
"""Full-size timing of the raw-pyramid training backward, piece by piece (dev tool; bench.py --mode train is the contract):
value_proj_heads_bwd, gather-dot, plan backward per layer; count per layer; scan + fill + reduce once per step."""
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
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--levels', default='r50')
    ap.add_argument('--no-walk', action='store_true')
    ap.add_argument('--regions', default='6,10')
    ap.add_argument('--alias', action='store_true', help='every camera row aliases row 0: the gather-dot without its L2 misses')
    a = ap.parse_args()
    dev = 'cuda'
    n, q, nl = 6 * a.frames, a.queries, a.layers
    levels = synthetic.R50_LEVELS if a.levels == 'r50' else synthetic.VOV_LEVELS
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in levels]
    l2i = torch.from_numpy(synthetic.camera_rig(a.frames)).unsqueeze(0).to(dev)
    sp, shapes = ops.pyramid_slice_planar_fwd(feats)
    pyr = ops.PyramidView.slice_planar(sp, shapes)
    if a.alias:
        pyr.cam_stride = [0] * len(pyr.cam_stride)
    del feats
    w_v = (torch.randn(256, 256, generator=g) / 16).to(dev)
    b_v = torch.randn(256, generator=g).to(dev)
    lay = []
    for _ in range(nl):
        ref = torch.rand(1, q, 3, generator=g).to(dev)
        offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * 1.5).to(dev)
        attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
        cam = torch.randn(1, q, n, generator=g).to(dev)
        order = ops.query_order_fwd(ref, synthetic.PC_RANGE)
        plan = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, query_order=order)
        lay.append((ref, offsets, attn, cam, plan, torch.randn(1, q, 256, generator=g).to(dev)))
    sink = ops.PyramidGrad(pyr, nl, 1, q, 8, chunk_walk=False)
    if not a.no_walk:
        sink.order = ops._chunk_walk(pyr.level_hw, pyr.rows, pyr.device, tuple(int(x) for x in a.regions.split(',')))
    dpart = torch.empty(ops.cross_attn_dot_bytes(1, n, q, 8), device=dev, dtype=torch.uint8)
    beta = torch.empty(1, q, 8, device=dev)
    status = torch.zeros(1, device=dev, dtype=torch.int32)

    def heads():
        for i, L in enumerate(lay):
            ops.value_proj_heads_bwd(L[5], w_v, b_v, 8, grad_agg=sink.grad_agg_rows(i), beta=beta)

    def dots():
        for i, L in enumerate(lay):
            ops.cross_attn_dot_sliced(L[4], sink.grad_agg_rows(i), dpart=dpart)

    def planb():
        for i, L in enumerate(lay):
            ops.cross_attn_plan_bwd(L[4], dpart, beta, L[0], L[1], L[2], L[3], l2i, synthetic.PC_RANGE, 900, 1600, status=status)

    def counts():
        sink.count.zero_()
        sink.plans = []
        for i, L in enumerate(lay):
            sink.add_layer(i, L[4])

    grads = [torch.empty(n, 256, h, w, device=dev) for h, w in levels]

    def finish():
        counts()
        sink.finish(grads)

    def prepare():
        counts()
        sink.prepare()

    heads()
    t_h = timed(heads, a.iters, nl)
    t_d = timed(dots, a.iters, nl)
    t_p = timed(planb, a.iters, nl)
    if a.alias:                                           # (the bookkeeping refuses aliased rows: the gather-dot's floor only)
        print(f'aliased camera rows: heads_bwd {t_h:.1f} us, gather-dot {t_d:.1f} us, plan_bwd {t_p:.1f} us per layer')
        return
    t_c = timed(counts, a.iters, 1)
    t_sf = timed(prepare, a.iters, 1)
    t_f = timed(finish, a.iters, 1)
    recs = int(sink.count.sum().item())
    print(f'per layer: heads_bwd {t_h:.1f} us, gather-dot {t_d:.1f} us, plan_bwd {t_p:.1f} us; per step ({nl} layers): zero + counts {t_c:.1f} us, '
          f'counts + scan + fill + sort {t_sf:.1f} us, all + reduce {t_f:.1f} us ({recs / 1e6:.2f} M records); status {int(status.item())}')


if __name__ == '__main__':
    main()
