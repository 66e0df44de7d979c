"""Place the slice-planar pyramid at the END of an allocator segment and run plan / gather / gather-dot on it: an over-read past
the pyramid then faults instead of going unnoticed (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_detr4d_amd import ops, synthetic

torch.manual_seed(0)
dev = 'cuda'
levels = [(8, 14), (4, 7), (2, 4), (1, 2)]
n, q, hh = 6, 32, 8
fd = [torch.randn(1, n, 256, h, w, device=dev) for h, w in levels]
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
sp0, hw = ops.pyramid_slice_planar_fwd(fd)
nbytes = sp0.numel() * 4
seg = ctypes.c_void_p()
SEG = 2 << 20
assert hip.hipMalloc(ctypes.byref(seg), ctypes.c_size_t(SEG)) == 0
base = seg.value + SEG - nbytes                         # the pyramid's last byte is the allocation's last byte
print('segment', hex(seg.value), 'pyramid at', hex(base), 'aligned end', (base + nbytes) % SEG == 0, flush=True)
torch.cuda.synchronize()
assert hip.hipMemcpy(ctypes.c_void_p(base), ctypes.c_void_p(sp0.data_ptr()), ctypes.c_size_t(nbytes), 3) == 0
starts = [0, 112, 140, 148]
pyr = ops.PyramidView([sp0], [base + st * 128 for st in starts], hw, [150 * 128] * 4, 128, n * 150 * 128, torch.float32, n)
l2i = torch.from_numpy(synthetic.camera_rig(1)).unsqueeze(0).to(dev)
ref = torch.rand(1, q, 3, device=dev)
off = torch.randn(1, q, hh, 4, 3, device=dev)
att = torch.randn(1, q, hh, 4, 4, device=dev)
cam = torch.randn(1, q, n, device=dev)
for both in (False, True):
    plan = ops.cross_attn_plan_fwd(pyr, ref, off, att, cam, l2i, synthetic.PC_RANGE, 64, 112, hh, both=both)
    torch.cuda.synchronize(); print('plan ok', both, flush=True)
    agg = ops.cross_attn_agg_sliced_fwd(plan)
    torch.cuda.synchronize(); print('agg ok', both, flush=True)
    gagg = torch.randn(1, q, hh, 256, device=dev)
    d = ops.cross_attn_dot_sliced(plan, gagg)
    torch.cuda.synchronize(); print('dot ok', both, flush=True)
items = ops.cross_attn_plan_fwd(pyr, ref, off, att, cam, l2i, synthetic.PC_RANGE, 64, 112, hh, items=True)
ops.cross_attn_agg_sliced_fwd(items)
torch.cuda.synchronize(); print('items agg ok', flush=True)
