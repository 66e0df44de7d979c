#!/usr/bin/env python
"""Per-shape GPU time of the small dense kernels, measured as hipGraph replays of many back-to-back
launches (removes the ~10 us host cost of an eager ctypes launch).  Dev tool."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def graph_time(fn, n=200):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
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
    for (m, k, n) in [(900, 256, 768), (900, 256, 256), (900, 256, 96), (900, 256, 24), (900, 256, 512),
                      (900, 512, 256), (900, 3, 256), (900, 256, 10)]:
        x, w, b = torch.randn(m, k, device='cuda'), torch.randn(n, k, device='cuda'), torch.randn(n, device='cuda')
        y = torch.empty(m, n, device='cuda')
        a = graph_time(lambda: ops.linear_fwd(x, w, b, out=y))
        t = graph_time(lambda: F.linear(x, w, b))
        print(f'linear {m}x{k}x{n}: gd4d {a:6.2f} us   torch {t:6.2f} us')
    for (m, k, n) in [(900, 256, 256), (900, 512, 256), (900, 256, 768), (900, 256, 512)]:
        x, w, b = torch.randn(m, k, device='cuda'), torch.randn(n, k, device='cuda'), torch.randn(n, device='cuda')
        r = torch.randn(m, n, device='cuda')
        ga, be = torch.randn(n, device='cuda'), torch.randn(n, device='cuda')
        if n == 256:
            a = graph_time(lambda: ops.linear_ln_fwd(x, w, b, ga, be, r1=r))
            t = graph_time(lambda: ops.layernorm_fwd(ops.linear_fwd(x, w, b, r1=r), ga, be))
            print(f'linear+LN {m}x{k}x{n}: row-block {a:6.2f} us   linear + layernorm launches {t:6.2f} us')
        a = graph_time(lambda: ops.linear_ln_fwd(x, w, b, r1=r))
        t = graph_time(lambda: ops.linear_fwd(x, w, b, r1=r))
        print(f'linear    {m}x{k}x{n}: row-block {a:6.2f} us   32x32-tile kernel {t:6.2f} us')
    x = torch.randn(900, 1, 256, device='cuda')
    g_, b_ = torch.randn(256, device='cuda'), torch.randn(256, device='cuda')
    print(f'layernorm 900x256: gd4d {graph_time(lambda: ops.layernorm_fwd(x, g_, b_)):6.2f} us   '
          f'torch {graph_time(lambda: F.layer_norm(x, (256,), g_, b_)):6.2f} us')
    qkv = torch.randn(900, 1, 768, device='cuda')
    q, k, v = qkv.split(256, dim=-1)
    print(f'mha core 900q: gd4d {graph_time(lambda: ops.mha_core_fwd(q, k, v, 8), 50):6.2f} us')
    tmp, ref = torch.randn(1, 900, 10, device='cuda'), torch.rand(1, 900, 3, device='cuda')
    print(f'refine: gd4d {graph_time(lambda: ops.refine_reference_fwd(tmp, ref)):6.2f} us')
