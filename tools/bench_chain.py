#!/usr/bin/env python
"""Standalone timing of gd4d_row_chain_fwd programs shaped like the decoder's chain A / chain B (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402


def timed(fn, iters=20):
    """us per call, `iters` calls captured into one hipGraph (eager launches of kernels this short are host-bound)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def main():
    dev, q, c = 'cuda', int(os.environ.get('Q', 900)), 256
    g = lambda *s: torch.randn(*s, device=dev) * 0.05  # noqa: E731
    ln = lambda n: torch.nn.LayerNorm(n).to(dev)       # noqa: E731
    x, pos, agg, x1, pf = g(q, c), g(q, c), g(q, c), g(q, c), g(q, c)
    w = {k: g(*s) for k, s in dict(o=(c, c), f1=(512, c), f2=(c, 512), inp=(3 * c, c), r1=(c, c), r2=(c, c), r3=(10, c),
                                   cam=(24, c), off=(96, c), att=(128, c)).items()}
    b = {k: g(v.shape[0]) for k, v in w.items()}
    n0, n1, n2 = ln(c), ln(c), ln(c)
    out, qkv, ref, nref = g(q, c), g(q, 3 * c), torch.rand(q, 3, device=dev), torch.empty(q, 3, device=dev)
    cam, off, att = g(q, 24), g(q, 96), g(q, 128)
    agg8, wsum8 = g(q, 8, c), torch.rand(q, 8, device=dev)
    progs = {
        'HEADGEMM only (to global)': [ops.chain_headgemm(agg8, wsum8, w['o'], b['o'], out=out)],
        'HEADGEMM + GEMM 256x256': [ops.chain_headgemm(agg8, wsum8, w['o'], b['o'], dst=0), ops.chain_gemm(0, w['o'], b['o'], out=out)],
        'LOAD + GEMM + GEMM': [ops.chain_load(0, x), ops.chain_gemm(0, w['o'], b['o'], dst=1), ops.chain_gemm(1, w['o'], b['o'], out=out)],
        'one GEMM 256x256 (load + gemm to global)': [ops.chain_load(0, x), ops.chain_gemm(0, w['o'], b['o'], out=out)],
        'in_proj (2 loads, 2 GEMMs, N = 768)': [ops.chain_load(0, x, pos), ops.chain_load(1, x),
                                                 ops.chain_gemm(0, w['inp'][:512], b['inp'][:512], out=qkv[:, :512]),
                                                 ops.chain_gemm(1, w['inp'][512:], b['inp'][512:], out=qkv[:, 512:])],
        'chain A': [ops.chain_load(0, agg), ops.chain_load(3, x), ops.chain_gemm(0, w['o'], b['o'], dst=1, res=3), ops.chain_layernorm(1, n0, dst=2, out=x1),
                    ops.chain_add(0, 2, c, add=pos), ops.chain_gemm(0, w['cam'], b['cam'], out=cam),
                    ops.chain_gemm(0, w['off'], b['off'], out=off), ops.chain_gemm(0, w['att'], b['att'], out=att)],
        'chain B': [ops.chain_load(0, agg), ops.chain_load(3, x1, pf), ops.chain_gemm(0, w['o'], b['o'], dst=1, res=3),
                    ops.chain_layernorm(1, n1, dst=2), ops.chain_gemm(2, w['f1'], b['f1'], dst=0, relu=True),
                    ops.chain_gemm(0, w['f2'], b['f2'], dst=1, res=2), ops.chain_layernorm(1, n2, dst=3, out=out),
                    ops.chain_add(0, 3, c, add=pos), ops.chain_gemm(0, w['inp'][:512], b['inp'][:512], out=qkv[:, :512]),
                    ops.chain_gemm(3, w['inp'][512:], b['inp'][512:], out=qkv[:, 512:]),
                    ops.chain_gemm(3, w['r1'], b['r1'], dst=1, relu=True), ops.chain_gemm(1, w['r2'], b['r2'], dst=2, relu=True),
                    ops.chain_gemm(2, w['r3'], b['r3'], dst=1), ops.chain_refine(1, ref, nref)],
    }
    for name, prog in progs.items():
        print(f'{name:45s} {timed(lambda: ops.row_chain_fwd(prog, q)):8.1f} us')
    lin = lambda: ops.linear_fwd(x, w['o'], b['o'])   # noqa: E731
    print(f'{"gd4d_linear_fwd 256x256 (for scale)":45s} {timed(lin):8.1f} us')


if __name__ == '__main__':
    main()
