#!/usr/bin/env python
"""Dev: where does the fused decoder loop (row chains, split-bf16 x3 products) differ from the oracle at the full size?
Layer 0 of the bench workload: the inputs of the gather (offsets / logits) and the layer output, module path vs fused path."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graph_detr4d_amd as G  # noqa: E402
from graph_detr4d_amd import functional as Fn, fused_decoder, synthetic  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def main():
    torch.set_num_threads(16)
    n, queries = 24, 900
    tr, regs = bench.build_decoder(G, n, 6, 'fp32', 1002)
    feats = synthetic.feature_pyramid(n, synthetic.R50_LEVELS, seed=78)
    qe = torch.randn(queries, 512, generator=torch.Generator().manual_seed(6))
    metas = synthetic.make_img_metas(synthetic.camera_rig(4), batch=1)
    sd, layer_params = bench.state_as_oracle_params(tr)
    pc = synthetic.PC_RANGE
    query_pos, query = (t.unsqueeze(1).contiguous() for t in torch.split(qe, 256, dim=1))
    ref0 = torch.nn.functional.linear(query_pos.permute(1, 0, 2), sd['reference_points.weight'], sd['reference_points.bias']).sigmoid()
    y_ref, parts = O.decoder_layer(layer_params[0], query, feats, query_pos, ref0, metas, pc, cross='Deform3DCrossAttn',
                                   num_heads=8, num_points=4, return_parts=True)
    dev = 'cuda'
    tr_d, regs_d = tr.to(dev), regs.to(dev)
    feats_d = [f.to(dev) for f in feats]
    cap = {}
    orig = Fn.LateValues.aggregate

    def spy(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=None, **vp):
        cap.update(offsets=offsets.detach().cpu().clone(), attn=attn_logits.detach().cpu().clone(), cam=cam_logits.detach().cpu().clone(),
                   ref=ref_.detach().cpu().clone())
        return orig(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order, **vp)
    Fn.LateValues.aggregate = spy
    layer = tr_d.decoder.layers[0]
    with torch.no_grad():
        for name in ('module', 'fused'):
            if name == 'module':
                y = layer(query.to(dev), key=None, value=feats_d, query_pos=query_pos.to(dev), reference_points=ref0.to(dev), img_metas=metas)
            else:
                late = Fn.LateValues(feats_d)
                out_all, _ = fused_decoder.run(types.SimpleNamespace(layers=[layer]), query.to(dev), query_pos.to(dev), feats_d, ref0.to(dev),
                                               [regs_d[0]], metas, None, None, None, None, pc, True, late=late)
                late.finish()
                y = out_all[0]
            e = (y.cpu() - y_ref).abs().amax(dim=(1, 2))
            print(f'{name:7s} layer output: max row err {e.max():.2e}  median {e.median():.2e}  rows > 1e-3: {(e > 1e-3).sum().item()}')
            for k, ref_t in (('offsets', parts['offsets']), ('attn', parts['attn_logits']), ('cam', parts['cam_logits'])):
                d = (cap[k].reshape(ref_t.shape) - ref_t).abs()
                print(f'        {k:8s} max abs err {d.max():.2e}  (max |value| {ref_t.abs().max():.2e})')
    Fn.LateValues.aggregate = orig


main()
