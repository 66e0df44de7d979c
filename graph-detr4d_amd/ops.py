"""Tensor-level wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function requires CUDA(HIP)-resident, contiguous tensors and raises otherwise - the product
path has no CPU fallback.
"""
import ctypes

import torch

from . import _lib


def _dev(t, name, dtype=None):
    if not t.is_cuda:
        raise _lib.Gd4dError(f'{name} must live on the GPU (no CPU fallback in graph-detr4d_amd)')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _value_dtype(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise TypeError(f'value tensors must be float32 or bfloat16, got {t.dtype}')


def cross_attn_fwd(value, level_hw, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                   img_h, img_w, want_mask=False, want_uv=False, out=None):
    """gd4d_cross_attn_fwd.  value (B*N, S, Hh, Dh); ref (B,Q,3); offsets (B,Q,Hh,P,3);
    attn_logits (B,Q,Hh,L,P) (or (B,Q,Hh,L*P)); cam_logits (B,Q,N); lidar2img (B,N,4,4).
    Returns out (B,Q,Hh*Dh) [, mask (B,N,Q,Hh,P) uint8] [, uv (B,N,Q,Hh,P,2)]."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    n = lidar2img.shape[1]
    hh, dh = value.shape[2], value.shape[3]
    p = offsets.shape[3]
    nl = len(level_hw)
    if value.shape[0] != b * n or value.shape[1] != sum(h * w for h, w in level_hw):
        raise ValueError(f'value shape {tuple(value.shape)} inconsistent with B*N={b * n}, '
                         f'levels {level_hw}')
    if attn_logits.numel() != b * q * hh * nl * p or cam_logits.numel() != b * q * n:
        raise ValueError('attn_logits / cam_logits have the wrong number of elements')
    if out is None:
        out = torch.empty(b, q, hh * dh, device=ref.device, dtype=torch.float32)
    mask = torch.empty(b, n, q, hh, p, device=ref.device, dtype=torch.uint8) if want_mask else None
    uv = torch.empty(b, n, q, hh, p, 2, device=ref.device, dtype=torch.float32) if want_uv else None
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    f32 = torch.float32
    code = lib.gd4d_cross_attn_fwd(
        _dev(value, 'value'), lv, _dev(ref, 'ref', f32), _dev(offsets, 'offsets', f32),
        _dev(attn_logits, 'attn_logits', f32), _dev(cam_logits, 'cam_logits', f32),
        _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w), _dev(out, 'out', f32),
        _dev(mask, 'mask') if want_mask else None, _dev(uv, 'uv') if want_uv else None,
        b, n, q, hh, dh, nl, p, _value_dtype(value), _stream())
    _lib.check(code, 'gd4d_cross_attn_fwd')
    res = (out,)
    if want_mask:
        res += (mask,)
    if want_uv:
        res += (uv,)
    return res if len(res) > 1 else out


def detr3d_fwd(feats, ref, attn_logits, lidar2img, pc_range, img_h, img_w, want_out=True,
               want_mask=False, want_sampled=False):
    """gd4d_detr3d_fwd.  feats: list of L tensors (B, N, C, H_l, W_l) fp32 (NCHW per camera);
    ref (B,Q,3); attn_logits (B,Q,N,1,L) (any shape with B*Q*N*L elements in that order).
    Returns a dict with the requested 'out' (B,Q,C), 'mask' (B,N,Q) uint8,
    'sampled' (B,C,Q,N,1,L)."""
    lib = _lib.load()
    b, n, c = feats[0].shape[:3]
    q = ref.shape[1]
    nl = len(feats)
    if attn_logits.numel() != b * q * n * nl:
        raise ValueError('attn_logits must have B*Q*N*L elements (num_points must be 1)')
    f32 = torch.float32
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    dev = ref.device
    out = torch.empty(b, q, c, device=dev, dtype=f32) if want_out else None
    mask = torch.empty(b, n, q, device=dev, dtype=torch.uint8) if want_mask else None
    sampled = torch.empty(b, c, q, n, 1, nl, device=dev, dtype=f32) if want_sampled else None
    code = lib.gd4d_detr3d_fwd(
        ptrs, lv, _dev(ref, 'ref', f32), _dev(attn_logits, 'attn_logits', f32),
        _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w),
        _dev(out, 'out') if want_out else None, _dev(mask, 'mask') if want_mask else None,
        _dev(sampled, 'sampled') if want_sampled else None, b, n, q, c, nl, 1, _stream())
    _lib.check(code, 'gd4d_detr3d_fwd')
    return dict(out=out, mask=mask, sampled=sampled)


def value_proj_fwd(feats, weight, bias, out_dtype=torch.float32, out=None):
    """gd4d_value_proj_fwd.  feats: list of L tensors (B, N, C, H_l, W_l) or (R, C, H_l, W_l) fp32;
    weight (C, C); bias (C) or None.  Returns (R, S, C) in `out_dtype`."""
    lib = _lib.load()
    f32 = torch.float32
    c = weight.shape[0]
    r = feats[0].numel() // (c * feats[0].shape[-1] * feats[0].shape[-2])
    nl = len(feats)
    s = sum(f.shape[-1] * f.shape[-2] for f in feats)
    if out is None:
        out = torch.empty(r, s, c, device=weight.device, dtype=out_dtype)
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    code = lib.gd4d_value_proj_fwd(ptrs, lv, _dev(weight, 'weight', f32),
                                   _dev(bias, 'bias', f32) if bias is not None else None,
                                   _dev(out, 'out'), r, c, nl, _lib.F32, _value_dtype(out), _stream())
    _lib.check(code, 'gd4d_value_proj_fwd')
    return out


def value_proj_multi_fwd(feats, weights, biases, out_dtype=torch.float32):
    """gd4d_value_proj_multi_fwd: project the same pyramid with NL (weight, bias) pairs in one
    launch.  Returns a list of NL tensors (R, S, C)."""
    lib = _lib.load()
    f32 = torch.float32
    nlayers = len(weights)
    c = weights[0].shape[0]
    r = feats[0].numel() // (c * feats[0].shape[-1] * feats[0].shape[-2])
    nl = len(feats)
    s = sum(f.shape[-1] * f.shape[-2] for f in feats)
    outs = [torch.empty(r, s, c, device=weights[0].device, dtype=out_dtype) for _ in range(nlayers)]
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    wp = (ctypes.c_void_p * nlayers)(*[_dev(w, 'weight', f32).value for w in weights])
    bp = (ctypes.c_void_p * nlayers)(*[(_dev(b, 'bias', f32).value if b is not None else None)
                                       for b in biases])
    op = (ctypes.c_void_p * nlayers)(*[_dev(o, 'out').value for o in outs])
    code = lib.gd4d_value_proj_multi_fwd(ptrs, lv, wp, bp, op, r, c, nl, nlayers, _lib.F32,
                                         _value_dtype(outs[0]), _stream())
    _lib.check(code, 'gd4d_value_proj_multi_fwd')
    return outs
