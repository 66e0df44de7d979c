"""Times the head position embedding's dense part at the bench's size (24 cameras, R50 pyramid): gd4d_mlp2_frustum_fwd +
gd4d_mlp2_se_fuse_fwd against gd4d_mlp2_pe_se_fwd (one kernel, the embedding kept in registers)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import graph_detr4d_amd as G  # noqa: F401,E402
from graph_detr4d_amd import ops, synthetic  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 24
levels = list(synthetic.R50_LEVELS)
torch.manual_seed(0)
rig = synthetic.camera_rig((r + 5) // 6)[:r].astype(np.float64)
i2l = torch.from_numpy(np.linalg.inv(rig).astype(np.float32)).cuda()
w1, b1 = (torch.randn(1024, 192) / 192 ** 0.5).cuda(), (torch.randn(1024) * 0.1).cuda()
w2, b2 = (torch.randn(256, 1024) / 32).cuda(), (torch.randn(256) * 0.1).cuda()
v1, c1 = (torch.randn(256, 256) / 16).cuda(), (torch.randn(256) * 0.1).cuda()
v2, c2 = (torch.randn(256, 256) / 16).cuda(), (torch.randn(256) * 0.1).cuda()
feats = [torch.randn(r, 256, h, w).cuda() for h, w in levels]
s_tot = sum(h * w for h, w in levels)
sine = torch.randn(r, s_tot, 256).cuda()
pe_img, se_img = ops.mlp2_frustum_image(w1, b1, w2), ops.mlp2_image(v1, c1, v2)
pe = torch.empty(r, s_tot, 256, device='cuda')
outs = [torch.empty(r, h, w, 256, device='cuda') for h, w in levels]
pad = (928, 1600)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def two():
    ops.mlp2_frustum_fwd(i2l, levels, pad, 64, 1.0, synthetic.PC_RANGE, pe_img, b2, out=pe)
    ops.mlp2_se_fuse_fwd(feats, se_img, c2, pe, sine)


for _ in range(2):
    print('cameras %d: frustum MLP %.3f ms, SE gate + fuse %.3f ms, both %.3f ms; one kernel %.3f ms' % (
        r, t(lambda: ops.mlp2_frustum_fwd(i2l, levels, pad, 64, 1.0, synthetic.PC_RANGE, pe_img, b2, out=pe)),
        t(lambda: ops.mlp2_se_fuse_fwd(feats, se_img, c2, pe, sine)), t(two),
        t(lambda: ops.mlp2_pe_se_fwd(i2l, feats, pad, 64, 1.0, synthetic.PC_RANGE, pe_img, b2, se_img, c2, sine, outs=outs))))
