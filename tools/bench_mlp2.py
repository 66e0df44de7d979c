#!/usr/bin/env python
"""Timing of gd4d_mlp2_bf16x3_fwd against the two gd4d_gemm_bf16x3_fwd launches it replaces (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 739800
    for k1, h in ((192, 1024), (256, 256)):
        torch.manual_seed(0)
        x = torch.randn(m, k1, device='cuda')
        w1, b1 = torch.randn(h, k1, device='cuda') / k1 ** 0.5, torch.randn(h, device='cuda')
        w2, b2 = torch.randn(256, h, device='cuda') / h ** 0.5, torch.randn(256, device='cuda')
        img = ops.mlp2_image(w1, b1, w2)
        out = torch.empty(m, 256, device='cuda')
        t_f = timed(lambda: ops.mlp2_bf16x3_fwd(x, img, b2, out=out))
        s1, s2 = ops.split_bf16_fwd(w1), ops.split_bf16_fwd(w2)
        hid = torch.empty(m, h, device='cuda')
        t_2 = timed(lambda: ops.gemm_bf16x3_fwd(ops.gemm_bf16x3_fwd(x, *s1, b1, relu=True, out=hid), *s2, b2, out=out))
        fl = 2.0 * m * (k1 * h + h * 256)
        print(f'M = {m}, {k1} -> {h} -> 256: fused {t_f:.3f} ms ({3 * fl / t_f / 1e9:.0f} TFLOP/s of bf16 products, {3 * fl / t_f / 1e9 / 2500 * 100:.0f} % of 2.5 PF), '
              f'two GEMMs {t_2:.3f} ms')
    # the SE form: NCHW levels in, gate + fuse in the epilogue, channels-last levels out (24 cameras, R50 level sizes)
    from graph_detr4d_amd import synthetic
    r = 24
    feats = [torch.randn(r, 256, h_, w_, device='cuda') for h_, w_ in synthetic.R50_LEVELS]
    s_tot = sum(f.shape[2] * f.shape[3] for f in feats)
    w1, b1 = torch.randn(256, 256, device='cuda') / 16, torch.randn(256, device='cuda')
    w2, b2 = torch.randn(256, 256, device='cuda') / 16, torch.randn(256, device='cuda')
    img = ops.mlp2_image(w1, b1, w2)
    pe, sine = torch.randn(r, s_tot, 256, device='cuda'), torch.randn(r, s_tot, 256, device='cuda')
    t_se = timed(lambda: ops.mlp2_se_fuse_fwd(feats, img, b2, pe, sine))
    s2 = ops.split_bf16_fwd(w2)

    def three():
        g1 = ops.value_proj_fwd([f.view(1, r, 256, f.shape[2], f.shape[3]) for f in feats], w1, b1)
        gate = ops.gemm_bf16x3_fwd(g1.view(r * s_tot, -1), *s2, b2, relu_in=True).view(r, s_tot, -1)
        st = 0
        for f in feats:
            ops.se_fuse_chlast_fwd(f, gate, pe, sine, st, out_channels_last=True)
            st += f.shape[2] * f.shape[3]
    t_3 = timed(three)
    print(f'SE gate + fuse over {r * s_tot} pixels: one kernel {t_se:.3f} ms, value_proj + GEMM + transposing fuse {t_3:.3f} ms')


if __name__ == '__main__':
    main()
