#!/usr/bin/env python
"""Device-side timeline of ONE replayed decoder step without a profiler (dev tool): block 0 of the row-chain, attention
core, channels-last copy and aggregate kernels stamps s_memrealtime (100 MHz) into a buffer (gd4d_trace_enable)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graph_detr4d_amd as G  # noqa: E402
from graph_detr4d_amd import _lib, synthetic  # noqa: E402

NAMES = {1: 'row_chain', 2: 'mha_core', 3: 'aggregate', 4: 'channels_last_copy', 5: 'plan', 6: 'sliced_aggregate',
         7: 'row_chain (second program)', 9: 'value_proj guest (first guest workgroup)'}


def main():
    dev = torch.device('cuda', 0)
    n_cams, layers, queries = 24, 6, 900
    tr, regs = bench.build_decoder(G, n_cams, layers, 'fp32', 1002)
    feats = [f.to(dev) for f in synthetic.feature_pyramid(n_cams, synthetic.R50_LEVELS, seed=1002)]
    if 'nhwc' in sys.argv[1:]:                           # the levels stored channels-last: gathered in place, no per-sample copy
        feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
        torch.cuda.synchronize()
    qe = torch.randn(queries, 512, generator=torch.Generator().manual_seed(1005)).to(dev)
    metas = synthetic.make_img_metas(synthetic.camera_rig(4), batch=1)
    tr, regs = tr.to(dev), regs.to(dev)
    step = lambda: tr(feats, qe, reg_branches=regs, img_metas=metas)   # noqa: E731
    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(graph):
            step()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        cap = 4096
        buf = torch.zeros(2 + 2 * cap, dtype=torch.int64, device=dev)
        buf[1] = cap
        lib = _lib.load()
        lib.gd4d_trace_enable(ctypes.c_void_p(buf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        lib.gd4d_trace_enable(None)
    n = int(buf[0].item())
    ent = buf[2:2 + 2 * n].view(n, 2).cpu().tolist()
    ent.sort(key=lambda e: e[1])
    t0 = ent[0][1]
    print(f'{n} stamps; step (events) {e0.elapsed_time(e1) * 1e3:.1f} us; first stamp -> last stamp {(ent[-1][1] - t0) / 100:.1f} us')
    prev = t0
    for ident, t in ent:
        if ident & 0x40 and not ident & 0x80:          # RC_TRACE_OPS build: end of one chain operation (block 0)
            second = bool(ident >> 63)
            ident &= (1 << 63) - 1
            opk, n_, k_ = (ident >> 8) & 0xff, (ident >> 16) & 0xffff, ident >> 32
            opn = {1: 'LOAD', 2: 'GEMM', 3: 'LAYERNORM', 4: 'ADD', 5: 'REFINE', 6: 'SMALL_LINEAR', 7: 'HEADGEMM'}.get(opk, str(opk))
            print(f'{(t - t0) / 100:9.1f} us  (+{(t - prev) / 100:7.1f})      {"[second program] " if second else ""}op {opn} N={n_} K={k_}')
            prev = t
            continue
        kind, end, nops = ident & 0x3f, bool(ident & 0x80), ident >> 8
        name = NAMES.get(kind, str(kind)) + (f'[{nops} ops]' if kind in (1, 7) else '')
        print(f'{(t - t0) / 100:9.1f} us  (+{(t - prev) / 100:7.1f})  {"end  " if end else "start"} {name}')
        prev = t


if __name__ == '__main__':
    main()
