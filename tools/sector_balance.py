"""Dev: how evenly do the 8 XCD sectors of the locality order share the aggregate kernel's work (visible tuples) on the
bench workload?  python tools/sector_balance.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import graph_detr4d_amd as G                           # noqa: E402
from graph_detr4d_amd import functional as Fn, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    tr, regs = bench.build_decoder(G, 24, 6, 'fp32', 1002)
    tr, regs = tr.to(dev), regs.to(dev)
    feats = [f.to(dev) for f in synthetic.feature_pyramid(24, synthetic.R50_LEVELS, seed=1002)]
    qe = torch.randn(900, 512, generator=torch.Generator().manual_seed(1005)).to(dev)
    metas = synthetic.make_img_metas(synthetic.camera_rig(4), batch=1)
    caps = []
    orig = Fn.LateValues.aggregate

    def spy(self, module, ref, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=None, **vp):
        from graph_detr4d_amd import ops
        mask = ops.cross_attn_agg_fwd(self.cl, self.shapes, ref.contiguous(), offsets.contiguous(), attn_logits.contiguous(),
                                      cam_logits.contiguous(), lidar2img, module.pc_range, img_h, img_w, module.num_heads,
                                      want_mask=True, query_order=order)[2]
        caps.append((mask, order))
        return orig(self, module, ref, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order, **vp)
    Fn.LateValues.aggregate = spy
    with torch.no_grad():
        tr(feats, qe, reg_branches=regs, img_metas=metas)
    Fn.LateValues.aggregate = orig
    for lid, (mask, order) in enumerate(caps):
        per_q = mask.sum(dim=(0, 1, 3, 4)).float()          # visible (camera, head, point) per query
        q = per_q.numel()
        per_xcd = (q + 7) // 8
        o = order.long() if order is not None else torch.arange(q, device=dev)
        sums = [per_q[o[x * per_xcd:(x + 1) * per_xcd]].sum().item() for x in range(8)]
        mean = sum(sums) / 8
        print(f'layer {lid}: visible tuples per sector {[int(s) for s in sums]}  max/mean {max(sums) / mean:.3f}  min/mean {min(sums) / mean:.3f}')


main()
