"""Time of gd4d_chain_weight_image_group for the bench's decoder (six layers + reg branches: ~160 jobs, 16 k fragments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_detr4d_amd as G
from graph_detr4d_amd import fused_train, synthetic
import bench

tr, regs = bench.build_decoder(G, 24, 6, 'fp32', 5)
tr, regs = tr.cuda(), regs.cuda()
imgs = fused_train._Images(tr.decoder, regs, torch.device('cuda', 0))
for _ in range(3):
    imgs.set.refresh()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    imgs.set.refresh()
b.record()
torch.cuda.synchronize()
print(f'{len(imgs.set._jobs)} jobs, {imgs.set._frags} fragments: {a.elapsed_time(b) / 50 * 1e3:.1f} us per launch')
