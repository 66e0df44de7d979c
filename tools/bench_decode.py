#!/usr/bin/env python
"""GPU time of the box decode (gd4d_nms_free_decode_fwd) and the head box epilogue as hipGraph replays, beside the
reference coder's chain of torch ops on the same GPU.  Dev tool; GD4D_LIB_PATH selects an ablation build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def graph_time(fn, n=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


with torch.no_grad():
    # the step after the decoder: one decode launch vs the reference coder's chain of torch ops on the same GPU
    POST = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]
    for qn in (900, 2700):
        cls = torch.randn(1, qn, 10, device='cuda') * 2 - 2
        box = torch.randn(1, qn, 10, device='cuda')
        post = torch.tensor(POST, device='cuda')

        def torch_decode():
            s, idx = cls[0].sigmoid().view(-1).topk(300)
            bb = box[0][idx // 10]
            out = torch.cat([bb[:, 0:1], bb[:, 1:2], bb[:, 4:5], bb[:, 2:3].exp(), bb[:, 3:4].exp(), bb[:, 5:6].exp(),
                             torch.atan2(bb[:, 6:7], bb[:, 7:8]), bb[:, 8:10]], -1)
            m = (out[:, :3] >= post[:3]).all(1) & (out[:, :3] <= post[3:]).all(1)
            return out, s, idx % 10, m
        print(f'nms-free decode {qn}q top-300: gd4d {graph_time(lambda: ops.nms_free_decode_fwd(cls, box, POST, 300), 50):6.2f} us'
              f'   torch ops {graph_time(torch_decode, 50):6.2f} us')
    tmp, ref = torch.randn(1, 900, 10, device='cuda'), torch.rand(1, 900, 3, device='cuda')
    print(f'box head 900q: gd4d {graph_time(lambda: ops.box_head_fwd(tmp, ref, POST)):6.2f} us')
