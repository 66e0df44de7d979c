#!/usr/bin/env python
"""gd4d_gemm_bf16x3_fwd against the library fp32 GEMM at the head position embedding's shapes.  Dev tool."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (m, k, n) in [(739800, 192, 1024), (739800, 1024, 256), (739800, 256, 256), (184950, 192, 1024)]:
    a = torch.randn(m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') * 0.05
    b = torch.randn(n, device='cuda')
    hi, lo = ops.split_bf16_fwd(w)
    out = torch.empty(m, n, device='cuda')
    t = timed(lambda: ops.gemm_bf16x3_fwd(a, hi, lo, b, relu=True, out=out))
    tl = timed(lambda: F.linear(a, w, b))
    fl = 2.0 * m * k * n
    print(f'M={m} K={k} N={n}: gd4d bf16x3 {t:.3f} ms ({fl / t / 1e9:.0f} TFLOP/s fp32-equivalent, x3 on the MFMA)   '
          f'library fp32 {tl:.3f} ms ({fl / tl / 1e9:.0f} TFLOP/s)')
