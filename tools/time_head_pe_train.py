"""Forward + backward of FeaturePositionEmbedding at the VoVNet / 24-camera size on both routes (GD4D_HEAD_PE_BWD=hip|torch):
milliseconds per call, and the largest difference between the two routes' gradients."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import FeaturePositionEmbedding  # noqa: E402


def main():
    torch.manual_seed(0)
    n = int(os.environ.get('CAMS', 24))
    shapes = [(116, 200), (58, 100), (29, 50), (15, 25)]
    pc = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    mod = FeaturePositionEmbedding(pc_range=pc).cuda()
    rng = np.random.default_rng(0)
    l2i = [np.eye(4) + 0.1 * rng.standard_normal((4, 4)) for _ in range(n)]
    metas = [dict(pad_shape=[(928, 1600, 3)] * n, img_shape=[(900, 1600, 3)] * n, lidar2img=l2i)]
    feats0 = [torch.randn(1, n, 256, h, w, device='cuda') for h, w in shapes]
    probes = [torch.randn_like(f) for f in feats0]
    res = {}
    for route in ('hip', 'torch', 'hip', 'torch'):
        os.environ['GD4D_HEAD_PE_BWD'] = route
        times = []
        for it in range(4):
            mod.zero_grad(set_to_none=True)
            feats = [f.clone().requires_grad_() for f in feats0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = mod(feats, metas)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            torch.autograd.backward(outs, probes)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            times.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        print(route, 'fwd / bwd ms:', ' '.join(f'{a:.1f}/{b:.1f}' for a, b in times),
              f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
        res[route] = ([f.grad.clone() for f in feats], {k: p.grad.clone() for k, p in mod.named_parameters()})
        torch.cuda.reset_peak_memory_stats()
    for a, b in zip(res['hip'][0], res['torch'][0]):
        print('feat grad max diff', (a - b).abs().max().item(), 'of', b.abs().max().item())
    for k in res['hip'][1]:
        a, b = res['hip'][1][k], res['torch'][1][k]
        print(k, 'max diff', (a - b).abs().max().item(), 'of', b.abs().max().item())


if __name__ == '__main__':
    main()
