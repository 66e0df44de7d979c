#!/usr/bin/env python
"""Quick full-size timing of gd4d_value_proj_(multi_)fwd (dev tool)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cams', type=int, default=24)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--out', default='f32')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--head-major', action='store_true')
    ap.add_argument('--bf16-math', action='store_true')
    ap.add_argument('--zeros', action='store_true', help='zero-filled operands (DVFS / power probe)')
    ap.add_argument('--max-cus', type=int, default=0)
    a = ap.parse_args()
    dev = 'cuda'
    feats = [torch.randn(a.cams, 256, h, w, device=dev) for h, w in synthetic.R50_LEVELS]
    ws = [torch.randn(256, 256, device=dev) * 0.06 for _ in range(a.layers)]
    ws_list = ws
    bs = [torch.randn(256, device=dev) for _ in range(a.layers)]
    if a.zeros:
        feats = [torch.zeros_like(f) for f in feats]
        ws = [torch.zeros_like(w) for w in ws]
    odt = torch.float32 if a.out == 'f32' else torch.bfloat16
    run = lambda: ops.value_proj_multi_fwd(feats, ws, bs, odt, head_major=a.head_major, bf16_math=a.bf16_math,  # noqa: E731
                                           max_cus=a.max_cus)
    outs = run()
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    rd = sum(f.numel() * 4 for f in feats)
    wr = sum(o.numel() * o.element_size() for o in outs)
    fl = 2 * 256 * 256 * sum(f.shape[0] * f.shape[2] * f.shape[3] for f in feats) * a.layers
    print(f'head_major={int(a.head_major)} cams={a.cams} layers={a.layers} out={a.out}: {ms * 1e3:.1f} us  '
          f'min-bytes {(rd + wr) / 1e6:.0f} MB -> {(rd + wr) / ms / 1e9:.2f} TB/s  '
          f'{fl / ms / 1e9:.0f} TFLOP/s (x3 MFMA work: {3 * fl / ms / 1e9:.0f})')


if __name__ == '__main__':
    main()
