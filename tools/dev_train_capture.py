#!/usr/bin/env python
"""Dev: forward + backward of the 2-layer golden decoder captured in a hipGraph (what tests/test_training_gpu.py does),
with switches to bisect a capture problem.  --no-feat-grad: frozen pyramid."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import graph_detr4d_amd as G  # noqa: E402
from golden_io import Golden  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-feat-grad', action='store_true')
    ap.add_argument('--no-capture', action='store_true')
    a = ap.parse_args()
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=m['num_layers'], return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                         embed_dims=256, dropout=0.0)],
                         feedforward_channels=512, ffn_dropout=0.0,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    tr.load_state_dict(g.state(), strict=True)
    tr = tr.to('cuda').eval()
    qe = g.t('query_embed').to('cuda')
    feats = [f.to('cuda').requires_grad_(not a.no_feat_grad) for f in g.feats()]
    metas = g.img_metas()

    def step():
        for p in tr.parameters():
            p.grad = None
        for f in feats:
            f.grad = None
        states, _, _ = tr(feats, qe, reg_branches=None, img_metas=metas)
        (states ** 2).mean().backward()

    step()
    torch.cuda.synchronize()
    print('eager ok', flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print('side ok', flush=True)
    if a.no_capture:
        return
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    print('captured', flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print('replayed', flush=True)


if __name__ == '__main__':
    main()
