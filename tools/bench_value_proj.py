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
    if int(os.environ.get('GD4D_VA_DBG', '0')) & 16:
        import ctypes
        from graph_detr4d_amd import _lib
        nb = _lib.load().gd4d_value_proj_workspace_bytes(a.layers)
        ws = torch.zeros(nb, device=dev, dtype=torch.uint8)
        f32 = torch.float32
        nl = len(feats)
        vp = ctypes.c_void_p
        ptrs = (vp * nl)(*[f.data_ptr() for f in feats])
        lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
        wp = (vp * a.layers)(*[w.data_ptr() for w in ws_list])
        bp = (vp * a.layers)(*[b.data_ptr() for b in bs])
        op = (vp * a.layers)(*[o.data_ptr() for o in outs])
        code = _lib.load().gd4d_value_proj_multi_fwd(ptrs, lv, wp, bp, op, a.cams, 256, nl, a.layers, 8, 0, 0, 0, 0,
                                                     vp(ws.data_ptr()), ctypes.c_size_t(nb), 0,
                                                     vp(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        tr = ws[nb - 4096:].view(torch.int64).view(-1, 8).cpu()
        for wv in range(16):
            t = tr[wv].tolist()
            n = max(t[3], 1)
            n = max(t[2], 1)
            if int(os.environ.get('GD4D_VA_DBG', '0')) & 32:
                print(f'  block {wv // 8} wave {wv % 8}: phases {t[2]}  piece steps {t[3] / n:7.0f}  store steps {t[1] / n:7.0f}  remaining steps {t[0] / n:7.0f}  cycles/phase')
            else:
                print(f'  block {wv // 8} wave {wv % 8}: phases {t[2]}  wait+barrier {t[0] / n:7.0f}  k-loop {t[1] / n:7.0f}  cycles/phase')
    print(f'head_major={int(a.head_major)} cams={a.cams} layers={a.layers} out={a.out}: {ms * 1e3:.1f} us  '
          f'min-bytes {(rd + wr) / 1e6:.0f} MB -> {(rd + wr) / ms / 1e9:.2f} TB/s  '
          f'{fl / ms / 1e9:.0f} TFLOP/s (x3 MFMA work: {3 * fl / ms / 1e9:.0f})')


if __name__ == '__main__':
    main()
