"""How many DISTINCT 128-byte corner lines would one load instruction of the gather fetch if its 8 line slots held the SAME
(camera, point, level, corner) of the 8 heads (the texture unit merges identical lines of one instruction), instead of 8 corners
of one head?  Inputs of tools/bench_sliced.py (900 x 24, offsets ~ N(0, 1.5 m)) and the bench's decoder layer 0 offsets scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_detr4d_amd import ops, synthetic

dev = 'cuda'
frames, q = 4, 900
n = 6 * frames
g = torch.Generator().manual_seed(0)
levels = synthetic.R50_LEVELS
feats = [torch.randn(1, n, 256, h, w, generator=g).to(dev) for h, w in levels]
sp, shapes = ops.pyramid_slice_planar_fwd(feats)
pyr = ops.PyramidView.slice_planar(sp, shapes)
l2i = torch.from_numpy(synthetic.camera_rig(frames)).unsqueeze(0).to(dev)
for scale in (1.5, 0.5, 3.0):
    ref = torch.rand(1, q, 3, generator=g).to(dev)
    offsets = (torch.randn(1, q, 8, 4, 3, generator=g) * scale).to(dev)
    attn = torch.randn(1, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(1, q, n, generator=g).to(dev)
    plan, mask, uv = ops.cross_attn_plan_fwd(pyr, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, 8, want_mask=True, want_uv=True)
    m = mask[0].bool()                                     # (N, Q, Hh, P)
    u, v = uv[0, ..., 0], uv[0, ..., 1]                    # normalised [0, 1]
    vis_per_head = m.sum(dim=(0, 3)).float()               # (Q, Hh)
    union = m.any(dim=2)                                   # (N, Q, P)
    print(f'offset scale {scale}: visible items per head {vis_per_head.mean():.2f}; union items per query {union.sum(dim=(0, 2)).float().mean():.2f} '
          f'(heads seeing a union item: {m.sum(dim=2)[union].float().mean():.2f} of 8)')
    out = []
    for (h, w) in levels:
        x0 = torch.floor(u * w - 0.5).long().clamp(0, w - 1)
        y0 = torch.floor(v * h - 0.5).long().clamp(0, h - 1)
        pix = (y0 * w + x0)                                # (N, Q, Hh, P)
        pix = torch.where(m, pix, torch.full_like(pix, -1))
        # 8 slots = 8 heads of one (camera, query, point)
        s = pix.permute(0, 1, 3, 2).reshape(-1, 8)         # (N*Q*P, 8)
        s = s[(s >= 0).any(dim=1)]
        srt = s.sort(dim=1).values
        distinct = ((srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)).sum(dim=1) + (srt[:, 0] >= 0).long()
        vis = (s >= 0).sum(dim=1)
        # all 32 (head, point) of a (camera, query)
        s2 = pix.reshape(n, q, 32)
        s2 = s2.reshape(-1, 32)
        s2 = s2[(s2 >= 0).any(dim=1)]
        srt2 = s2.sort(dim=1).values
        d2 = ((srt2[:, 1:] != srt2[:, :-1]) & (srt2[:, 1:] >= 0)).sum(dim=1) + (srt2[:, 0] >= 0).long()
        out.append((float(distinct.sum()) / float(vis.sum()), float(d2.sum()) / float((s2 >= 0).sum())))
    print('   distinct / visible per level, slots = 8 heads of one (camera, point):', ' '.join(f'{a:.2f}' for a, _ in out),
          '| all 32 (head, point) of a (camera, query):', ' '.join(f'{b:.2f}' for _, b in out), flush=True)
