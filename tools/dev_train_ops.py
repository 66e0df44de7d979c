#!/usr/bin/env python
"""Dev: which ATen ops a training step of the 6-layer decoder still launches (torch profiler, one eager step)."""
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import graph_detr4d_amd as G  # noqa: E402
from graph_detr4d_amd import dist as D, synthetic  # noqa: E402


def main():
    dev = 'cuda'
    n_cams = 24
    tr, regs = bench.build_decoder(G, n_cams, 6, 'fp32', 1002)
    tr, regs = tr.to(dev), regs.to(dev)
    feats = [f.to(dev).requires_grad_(True) for f in synthetic.feature_pyramid(n_cams, synthetic.R50_LEVELS, seed=1)]
    qe = torch.randn(900, 512).to(dev)
    metas = synthetic.make_img_metas(synthetic.camera_rig(4), batch=1)
    params = list(tr.parameters()) + list(regs.parameters())
    red = D.FlatGradAllReducer(params)
    red.bind(fuse_weight_grads=True)
    tr.eval()

    def step():
        red.zero_grad()
        states, _, _ = tr(feats, qe, reg_branches=regs, img_metas=metas)
        (states ** 2).mean().backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        if e.key.startswith('aten::') and e.count > 0:
            rows.append((e.count, e.key))
    for c, k in sorted(rows, reverse=True)[:40]:
        print(f'{c:5d}  {k}')
    seen = set()
    for e in prof.events():
        if e.name in ('aten::clone', 'aten::copy_', 'aten::add') and e.input_shapes and e.input_shapes[0] and e.input_shapes[0][-1:] in ([512], [256]):
            st, par = [], e.cpu_parent
            while par is not None and len(st) < 4:
                st.append(par.name)
                par = par.cpu_parent
            st = tuple(st)
            key = (e.name, str(e.input_shapes[0]), st)
            if key not in seen:
                seen.add(key)
                print(e.name, e.input_shapes[0], st)
    print('--- by input shape')
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        if e.key in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::cat', 'aten::sum',
                     'aten::clone', 'aten::contiguous', 'aten::zeros_like', 'aten::zeros', 'aten::sub', 'aten::sigmoid'):
            rows.append((e.count, e.key, str(e.input_shapes)[:150]))
    for c, k, sh in sorted(rows, reverse=True)[:60]:
        print(f'{c:5d}  {k:16s} {sh}')


if __name__ == '__main__':
    main()
