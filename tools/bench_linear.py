#!/usr/bin/env python
"""Per-shape timing of gd4d_linear_fwd vs torch F.linear (dev tool)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def t(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (m, k, n) in [(900, 256, 768), (900, 256, 256), (900, 256, 128), (900, 256, 96), (900, 256, 24),
                  (900, 256, 512), (900, 512, 256), (900, 3, 256), (900, 256, 10)]:
    x, w, b = torch.randn(m, k, device='cuda'), torch.randn(n, k, device='cuda'), torch.randn(n, device='cuda')
    print(f'{m}x{k}x{n}: gd4d {t(lambda: ops.linear_fwd(x, w, b)):6.1f} us   torch {t(lambda: F.linear(x, w, b)):6.1f} us')
