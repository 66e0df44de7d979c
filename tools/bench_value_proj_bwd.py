#!/usr/bin/env python
"""Full-size timing of gd4d_value_proj_bwd_input / _bwd_weight against the library GEMMs autograd would run (dev tool)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cams', type=int, default=24)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--library', action='store_true', help='also time the torch.matmul / einsum backward')
    a = ap.parse_args()
    dev = 'cuda'
    levels = list(synthetic.R50_LEVELS)
    feats = [torch.randn(a.cams, 256, h, w, device=dev) for h, w in levels]
    w = torch.randn(256, 256, device=dev) * 0.06
    s = sum(h * ww for h, ww in levels)
    gout = torch.randn(a.cams, s, 256, device=dev) * 0.01
    nbytes = gout.numel() * 4
    grads = [torch.zeros_like(f) for f in feats]
    t = timed(lambda: ops.value_proj_bwd_input(gout, w, levels, grads=grads), a.iters)
    print(f'bwd_input  (overwrite): {t:8.1f} us   {2 * nbytes / t / 1e6:.2f} TB/s of read + write')
    t = timed(lambda: ops.value_proj_bwd_input(gout, w, levels, grads=grads, accumulate=True), a.iters)
    print(f'bwd_input (accumulate): {t:8.1f} us   {3 * nbytes / t / 1e6:.2f} TB/s of 2 reads + write')
    t = timed(lambda: ops.value_proj_bwd_weight(gout, feats), a.iters)
    print(f'bwd_weight            : {t:8.1f} us   {2 * nbytes / t / 1e6:.2f} TB/s of 2 reads')
    if a.library:
        flat = torch.cat([f.reshape(a.cams, 256, -1) for f in feats], 2)
        t = timed(lambda: torch.einsum('rso,rcs->oc', gout, flat), a.iters)
        print(f'library dW (einsum, pyramid already concatenated): {t:8.1f} us')
        t = timed(lambda: torch.matmul(gout, w).transpose(1, 2).contiguous(), a.iters)
        print(f'library dX (matmul + transposed copy)            : {t:8.1f} us')


if __name__ == '__main__':
    main()
