#!/usr/bin/env python
"""Per-launch GPU time of the small backward kernels of a training step (hipGraph replays of back-to-back launches).  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def graph_time(fn, n=50):
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


def main():
    dev = 'cuda'
    l, b, h, c = 900, 1, 8, 256
    qkv = torch.randn(l, b, 3 * c, device=dev)
    q, k, v = qkv.split(c, dim=-1)
    do = torch.randn(l, b, c, device=dev)
    mask = torch.zeros(l, l, dtype=torch.bool, device=dev)
    mask[300:, :300] = True
    mask[:300, 300:] = True
    seed = ops.mha_dropout_seed(dev)
    for name, m, p in (('none', None, 0.), ('bool mask', mask, 0.), ('dropout 0.1', None, 0.1)):
        out, lse = ops.mha_core_fwd(q, k, v, h, m, want_lse=True, dropout_p=p, seed=seed if p else None)
        f = graph_time(lambda: ops.mha_core_fwd(q, k, v, h, m, want_lse=True, dropout_p=p, seed=seed if p else None), 50)
        t = graph_time(lambda: ops.mha_core_bwd(q, k, v, out, do, lse, h, m, dropout_p=p, seed=seed if p else None), 50)
        print(f'mha core 900 x 900, {name}: forward {f:6.1f} us, backward (two kernels) {t:6.1f} us')
    for (k_, n_) in [(256, 512), (256, 256), (256, 1024), (1024, 256), (256, 128), (256, 96), (256, 24), (3, 256), (256, 10)]:
        xx = torch.randn(l, k_, device=dev)
        ww = torch.randn(n_, k_, device=dev)
        bb = torch.randn(n_, device=dev)
        gy = torch.randn(l, n_, device=dev)
        f = graph_time(lambda: ops.linear_fwd(xx, ww, bb), 50)
        t = graph_time(lambda: ops.linear_fwd(gy, ww, weight_kn=True), 50)
        w_ = graph_time(lambda: ops.linear_bwd_weight(xx, gy, want_bias=True), 50)
        print(f'linear 900 x {k_} -> {n_}: forward {f:5.1f} us, input gradient {t:5.1f} us, weight gradient {w_:5.1f} us')
    x = torch.randn(l, c, device=dev)
    g_, b_ = torch.randn(c, device=dev), torch.randn(c, device=dev)
    dy = torch.randn(l, c, device=dev)
    print(f'layernorm_bwd 900 x 256: {graph_time(lambda: ops.layernorm_bwd(x, g_, b_, dy), 50):6.1f} us')


if __name__ == '__main__':
    main()
