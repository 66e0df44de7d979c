"""Tensor-level wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function requires CUDA(HIP)-resident, contiguous tensors and raises otherwise - the product
path has no CPU fallback.
"""
import ctypes
import functools
import inspect
import os

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
                   img_h, img_w, want_mask=False, want_uv=False, out=None, head_major=False, query_order=None,
                   raw_cam_weights=False):
    """gd4d_cross_attn_fwd.  value (B*N, S, Hh, Dh), or (B*N, Hh, S, Dh) with head_major=True;
    ref (B,Q,3); offsets (B,Q,Hh,P,3);
    attn_logits (B,Q,Hh,L,P) (or (B,Q,Hh,L*P)); cam_logits (B,Q,N); lidar2img (B,N,4,4).
    query_order: optional int32 permutation of [0, B*Q) from query_order_fwd (scheduling only, same result).
    Returns out (B,Q,Hh*Dh) [, mask (B,N,Q,Hh,P) uint8] [, uv (B,N,Q,Hh,P,2)]."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    n = lidar2img.shape[1]
    hh, dh = (value.shape[1], value.shape[3]) if head_major else (value.shape[2], value.shape[3])
    p = offsets.shape[3]
    nl = len(level_hw)
    if value.shape[0] != b * n or value.shape[2 if head_major else 1] != sum(h * w for h, w in level_hw):
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
        b, n, q, hh, dh, nl, p, _value_dtype(value), _lib.HEAD_MAJOR if head_major else _lib.PIXEL_MAJOR,
        1 if raw_cam_weights else 0, None if query_order is None else _order_ptr(query_order, b * q), _stream())
    _lib.check(code, 'gd4d_cross_attn_fwd')
    res = (out,)
    if want_mask:
        res += (mask,)
    if want_uv:
        res += (uv,)
    return res if len(res) > 1 else out


def pyramid_channels_last_fwd(feats, out=None, max_cus=0, out_dtype=torch.float32):
    """gd4d_pyramid_channels_last_fwd.  feats: list of L tensors (R, 256, H_l, W_l) fp32 (or (B, N, 256, H, W)).
    max_cus > 0: one persistent workgroup on each of that many compute units (the rest stays free for another stream).
    out_dtype torch.bfloat16: bf16 storage of the copy (reduced precision; the aggregate kernel accumulates in fp32).
    Returns (cl (R, S, 256), level_hw)."""
    lib = _lib.load()
    fl = [f.reshape(-1, *f.shape[-3:]) for f in feats]
    r, c = fl[0].shape[0], fl[0].shape[1]
    level_hw = [(int(f.shape[-2]), int(f.shape[-1])) for f in fl]
    s = sum(h * w for h, w in level_hw)
    if any(f.shape[0] != r or f.shape[1] != c for f in fl):
        raise ValueError('feature levels disagree in rows / channels')
    if out is None:
        out = torch.empty(r, s, c, device=fl[0].device, dtype=out_dtype)
    ptrs = (ctypes.c_void_p * len(fl))(*[_dev(f, 'feats', torch.float32).value for f in fl])
    lv = (ctypes.c_int32 * (2 * len(fl)))(*[int(x) for hw in level_hw for x in hw])
    code = lib.gd4d_pyramid_channels_last_fwd(ptrs, lv, _dev(out, 'out'), r, c, len(fl), _lib.F32, _value_dtype(out), int(max_cus),
                                              _stream())
    _lib.check(code, 'gd4d_pyramid_channels_last_fwd')
    return out, level_hw


def cross_attn_agg_fwd(feats_cl, level_hw, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w,
                       num_heads, want_mask=False, want_uv=False, query_order=None, raw_cam_weights=False,
                       vp_weight=None, vp_bias=None):
    """gd4d_cross_attn_agg_fwd.  feats_cl (B*N, S, 256) fp32 channels-last pyramid; the other arguments as cross_attn_fwd.
    Returns agg (B, Q, Hh, 256), wsum (B, Q, Hh) [, mask] [, uv]; with vp_weight (256, 256) [, vp_bias]: value_proj of the
    aggregates applied in the kernel's epilogue - returns out (B, Q, 256) [, mask] [, uv] instead."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    n = lidar2img.shape[1]
    hh, p = num_heads, offsets.shape[3]
    nl = len(level_hw)
    c = feats_cl.shape[-1]
    if feats_cl.shape[0] != b * n or feats_cl.shape[1] != sum(h * w for h, w in level_hw):
        raise ValueError(f'feats_cl shape {tuple(feats_cl.shape)} inconsistent with B*N={b * n}, levels {level_hw}')
    if offsets.numel() != b * q * hh * p * 3 or attn_logits.numel() != b * q * hh * nl * p or cam_logits.numel() != b * q * n:
        raise ValueError('offsets / attn_logits / cam_logits have the wrong number of elements')
    f32 = torch.float32
    fused = vp_weight is not None
    agg = None if fused else torch.empty(b, q, hh, c, device=ref.device, dtype=f32)
    wsum = None if fused else torch.empty(b, q, hh, device=ref.device, dtype=f32)
    out = torch.empty(b, q, c, device=ref.device, dtype=f32) if fused else None
    mask = torch.empty(b, n, q, hh, p, device=ref.device, dtype=torch.uint8) if want_mask else None
    uv = torch.empty(b, n, q, hh, p, 2, device=ref.device, dtype=f32) if want_uv else None
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    code = lib.gd4d_cross_attn_agg_fwd(
        _dev(feats_cl, 'feats_cl'), lv, _dev(ref, 'ref', f32), _dev(offsets, 'offsets', f32),
        _dev(attn_logits, 'attn_logits', f32), _dev(cam_logits, 'cam_logits', f32), _dev(lidar2img, 'lidar2img', f32),
        rng, float(img_h), float(img_w), None if fused else _dev(agg, 'agg'), None if fused else _dev(wsum, 'wsum'),
        _dev(mask, 'mask') if want_mask else None, _dev(uv, 'uv') if want_uv else None,
        b, n, q, hh, c, nl, p, _value_dtype(feats_cl), 1 if raw_cam_weights else 0,
        None if query_order is None else _order_ptr(query_order, b * q),
        _dev(vp_weight, 'vp_weight', f32) if fused else None, _opt(vp_bias, 'vp_bias') if fused else None,
        _dev(out, 'out') if fused else None, _stream())
    _lib.check(code, 'gd4d_cross_attn_agg_fwd')
    res = (out,) if fused else (agg, wsum)
    if want_mask:
        res += (mask,)
    if want_uv:
        res += (uv,)
    return res


def pyramid_slice_planar_fwd(feats, out=None, max_cus=0, out_dtype=torch.float32):
    """gd4d_pyramid_slice_planar_fwd.  feats as pyramid_channels_last_fwd.  Returns (sp (8, R, S, 32), level_hw): the
    channels-last pyramid with the channel axis cut into 8 planes of 32 (the layout gd4d_cross_attn_agg_sliced_fwd
    gathers from when it has to make its own copy)."""
    lib = _lib.load()
    fl = [f.reshape(-1, *f.shape[-3:]) for f in feats]
    r, c = fl[0].shape[0], fl[0].shape[1]
    level_hw = [(int(f.shape[-2]), int(f.shape[-1])) for f in fl]
    s = sum(h * w for h, w in level_hw)
    if any(f.shape[0] != r or f.shape[1] != c for f in fl):
        raise ValueError('feature levels disagree in rows / channels')
    if c != 256:
        raise ValueError('the slice-planar copy is built for 256 channels')
    if out is None:
        out = torch.empty(8, r, s, 32, device=fl[0].device, dtype=out_dtype)
    ptrs = (ctypes.c_void_p * len(fl))(*[_dev(f, 'feats', torch.float32).value for f in fl])
    lv = (ctypes.c_int32 * (2 * len(fl)))(*[int(x) for hw in level_hw for x in hw])
    code = lib.gd4d_pyramid_slice_planar_fwd(ptrs, lv, _dev(out, 'out'), r, c, len(fl), _lib.F32, _value_dtype(out), int(max_cus),
                                             _stream())
    _lib.check(code, 'gd4d_pyramid_slice_planar_fwd')
    return out, level_hw


class PyramidView:
    """How gd4d_cross_attn_agg_sliced_fwd addresses a 256-channel pyramid: per-level base pointers + byte strides
    (include/gd4d.h).  Three sources: the slice-planar copy, the pixel-major copy, caller-owned channels-last levels
    read in place."""

    def __init__(self, tensors, ptrs, level_hw, cam_stride, pix_stride, slice_stride, dtype, rows):
        self.tensors, self.ptrs, self.level_hw = tensors, ptrs, [tuple(int(x) for x in hw) for hw in level_hw]
        self.cam_stride, self.pix_stride, self.slice_stride = [int(x) for x in cam_stride], int(pix_stride), int(slice_stride)
        self.dtype, self.rows = dtype, int(rows)

    @property
    def device(self):
        return self.tensors[0].device

    @staticmethod
    def _starts(level_hw):
        out, s = [], 0
        for h, w in level_hw:
            out.append(s)
            s += h * w
        return out, s

    @classmethod
    def slice_planar(cls, sp, level_hw):
        """sp (8, R, S, 32) from pyramid_slice_planar_fwd."""
        es = sp.element_size()
        starts, s = cls._starts(level_hw)
        if sp.dim() != 4 or sp.shape[0] != 8 or sp.shape[2] != s or sp.shape[3] != 32 or not sp.is_contiguous():
            raise ValueError(f'slice-planar pyramid {tuple(sp.shape)} inconsistent with levels {level_hw}')
        r = sp.shape[1]
        return cls([sp], [sp.data_ptr() + st * 32 * es for st in starts], level_hw, [s * 32 * es] * len(starts), 32 * es,
                   r * s * 32 * es, sp.dtype, r)

    @classmethod
    def pixel_major(cls, cl, level_hw):
        """cl (R, S, 256) from pyramid_channels_last_fwd."""
        es = cl.element_size()
        starts, s = cls._starts(level_hw)
        if cl.dim() != 3 or cl.shape[1] != s or cl.shape[2] != 256 or not cl.is_contiguous():
            raise ValueError(f'channels-last pyramid {tuple(cl.shape)} inconsistent with levels {level_hw}')
        return cls([cl], [cl.data_ptr() + st * 256 * es for st in starts], level_hw, [s * 256 * es] * len(starts), 256 * es,
                   32 * es, cl.dtype, cl.shape[0])

    @staticmethod
    def is_channels_last_level(t):
        """(..., 256, H, W) whose memory is (..., H, W, 256): what `x.permute(.., 2, 3, 1).contiguous().permute(.., 3, 1, 2)`
        or torch.channels_last on the (rows, C, H, W) view gives."""
        if t.dim() not in (4, 5) or t.shape[-3] != 256:
            return False
        c, h, w = t.shape[-3:]
        want = {-3: 1, -1: c, -2: w * c}
        if t.dim() == 5:
            want[1], want[0] = h * w * c, t.shape[1] * h * w * c
        else:
            want[0] = h * w * c
        return all(t.shape[d] == 1 or t.stride(d) == st for d, st in want.items())   # (the stride of a size-1 dim is free)

    @classmethod
    def channels_last_levels(cls, levels):
        """levels: L tensors (B, N, 256, H, W) or (R, 256, H, W) with channels-last strides, fp32 or bf16: no copy."""
        if not all(cls.is_channels_last_level(t) for t in levels) or len({t.dtype for t in levels}) != 1:
            raise ValueError('channels_last_levels needs (.., 256, H, W) tensors stored as (.., H, W, 256), one dtype')
        es = levels[0].element_size()
        rows = levels[0].numel() // (256 * levels[0].shape[-1] * levels[0].shape[-2])
        hw = [(int(t.shape[-2]), int(t.shape[-1])) for t in levels]
        return cls(list(levels), [t.data_ptr() for t in levels], hw, [h * w * 256 * es for h, w in hw], 256 * es, 32 * es,
                   levels[0].dtype, rows)


def cross_attn_plan_bytes(b, n, q, num_heads, points=4):
    return int(_lib.load().gd4d_cross_attn_plan_bytes(b, n, q, num_heads, points))


class Plan:
    """Output of cross_attn_plan_fwd: the plan buffer, the locality order it is stored by, the pyramid it addresses and
    wsum (B, Q, Hh) - the sum of the in-bounds sampling weights per head (items form: filled by the gather, not by the
    plan kernel).  items: the buffer holds the ITEMS form (include/gd4d.h, GD4D_CA_PLAN_ITEMS) - gather only; the
    training backward kernels need the pairs form."""

    def __init__(self, buf, order, pyramid, b, q, num_heads, wsum, items=False, points=4, items_buf=None):
        self.buf, self.order, self.pyramid, self.b, self.q, self.num_heads, self.wsum = buf, order, pyramid, b, q, num_heads, wsum
        self.items, self.points = bool(items), int(points)
        self.items_buf = items_buf           # both=True: `buf` holds the pairs, this view the items (the forward gather's form)

    def need_pairs(self, who):
        if self.items:
            raise _lib.Gd4dError(f'{who} reads the pairs form of the plan; this one was made with items=True')
        if self.points not in (4, 8) or (self.points == 8 and self.num_heads != 8):
            raise _lib.Gd4dError(f'{who} is built for 4 points per head (every shipped config) and for 8 with 8 heads; this plan has '
                                 f'{self.points} - functional.pad_points pads other counts up to them')


CA_RAW_CAM_WEIGHTS, CA_PLAN_ITEMS, CA_PLAN_BOTH = 1, 2, 16


def cross_attn_plan_fwd(pyramid, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, num_heads,
                        want_mask=False, want_uv=False, raw_cam_weights=False, plan=None, query_order=None, items=False,
                        both=False):
    """gd4d_cross_attn_plan_fwd: projection + mask + softmax + camera weights + bilinear corners of one decoder layer's
    cross-attention -> what gd4d_cross_attn_agg_sliced_fwd walks on `pyramid` (a PyramidView).  The other arguments as
    cross_attn_fwd; plan: a Plan to overwrite; items: the 32-bytes-per-item form (the corners are worked out by the gather,
    which then also fills wsum).  both: pairs AND items in one launch (a training step: the forward gather reads the items, the
    backward kernels the pairs).
    Returns Plan [, mask (B, N, Q, Hh, P) uint8] [, uv (B, N, Q, Hh, P, 2)]."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    n = lidar2img.shape[1]
    hh, p, nl = num_heads, offsets.shape[3], len(pyramid.level_hw)
    if pyramid.rows != b * n:
        raise ValueError(f'pyramid has {pyramid.rows} camera rows, expected B*N = {b * n}')
    if offsets.numel() != b * q * hh * p * 3 or attn_logits.numel() != b * q * hh * nl * p or cam_logits.numel() != b * q * n:
        raise ValueError('offsets / attn_logits / cam_logits have the wrong number of elements')
    f32 = torch.float32
    nbytes = cross_attn_plan_bytes(b, n, q, hh, p)
    if both and (items or plan is not None):
        raise ValueError('both=True excludes items / plan')
    buf = torch.empty(2 * nbytes if both else nbytes, device=ref.device, dtype=torch.uint8) if plan is None else plan.buf
    wsum = torch.empty(b, q, hh, device=ref.device, dtype=f32) if plan is None else plan.wsum
    plan = Plan(buf, query_order, pyramid, b, q, hh, wsum, items=items, points=p, items_buf=buf[nbytes:] if both else None)
    mask = torch.empty(b, n, q, hh, p, device=ref.device, dtype=torch.uint8) if want_mask else None
    uv = torch.empty(b, n, q, hh, p, 2, device=ref.device, dtype=f32) if want_uv else None
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in pyramid.level_hw for x in hw])
    cs = (ctypes.c_int64 * nl)(*pyramid.cam_stride)
    code = lib.gd4d_cross_attn_plan_fwd(
        _dev(ref, 'ref', f32), _dev(offsets, 'offsets', f32), _dev(attn_logits, 'attn_logits', f32),
        _dev(cam_logits, 'cam_logits', f32), _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w),
        lv, cs, pyramid.pix_stride, _dev(buf, 'plan', torch.uint8), buf.numel(), _dev(wsum, 'wsum', f32),
        _dev(mask, 'mask') if want_mask else None, _dev(uv, 'uv') if want_uv else None, b, n, q, hh, nl, p,
        (CA_RAW_CAM_WEIGHTS if raw_cam_weights else 0) | (CA_PLAN_ITEMS if items else 0) | (CA_PLAN_BOTH if both else 0),
        None if query_order is None else _order_ptr(query_order, b * q), _stream())
    _lib.check(code, 'gd4d_cross_attn_plan_fwd')
    res = (plan,)
    if want_mask:
        res += (mask,)
    if want_uv:
        res += (uv,)
    return res if len(res) > 1 else plan


def cross_attn_agg_sliced_fwd(plan, slices=(0, 8), agg=None, count=None):
    """gd4d_cross_attn_agg_sliced_fwd on the pyramid the Plan was made for.  Returns agg (B, Q, Hh, 256); with plan.wsum
    (B, Q, Hh) that is what cross_attn_agg_fwd returns (other summation order).
    count = (PyramidGrad, layer): a training step's forward - the records of this plan get their slots in the SAME launch
    (gd4d_cross_attn_agg_items_count_fwd; needs the plan in both forms, 8 heads, 4 levels, fp32; PyramidGrad.add_layer otherwise)."""
    lib = _lib.load()
    pyramid = plan.pyramid
    dev = pyramid.device
    b, q, hh, query_order = plan.b, plan.q, plan.num_heads, plan.order
    if not plan.buf.is_cuda or plan.buf.device != dev:
        raise _lib.Gd4dError('plan must live on the pyramid\'s GPU (no CPU fallback in graph-detr4d_amd)')
    nl = len(pyramid.level_hw)
    f32 = torch.float32
    if agg is None:
        agg = torch.empty(b, q, hh, 256, device=dev, dtype=f32)
    ptrs = (ctypes.c_void_p * nl)(*pyramid.ptrs)
    if plan.items or plan.items_buf is not None:
        lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in pyramid.level_hw for x in hw])
        cs = (ctypes.c_int64 * nl)(*pyramid.cam_stride)
        if count is not None:
            sink, layer = count
            if plan.items or plan.items_buf is None or tuple(slices) != (0, 8):
                raise _lib.Gd4dError('gather + record count in one launch: a plan in both forms, all slices')
            slots, slot_bytes = sink.begin_layer(layer, plan)
            code = lib.gd4d_cross_attn_agg_items_count_fwd(
                ptrs, lv, cs, pyramid.pix_stride, pyramid.slice_stride, _dev(plan.items_buf, 'plan', torch.uint8), _dev(agg, 'agg', f32),
                _dev(plan.wsum, 'wsum', f32), b, pyramid.rows // b, q, hh, 256, nl, plan.points,
                _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16,
                None if query_order is None else _order_ptr(query_order, b * q), _dev(plan.buf, 'plan', torch.uint8),
                _dev(sink.count, 'count', torch.int32), _dev(slots, 'slots'), ctypes.c_size_t(slot_bytes), _stream())
            if code == -2:                               # GD4D_EUNSUPPORTED (e.g. a pyramid of 4 GiB or more): the caller launches the two
                return None
            _lib.check(code, 'gd4d_cross_attn_agg_items_count_fwd')
            sink.plans.append((int(layer), plan, slots))
            return agg
        items_buf = _dev(plan.buf if plan.items else plan.items_buf, 'plan', torch.uint8)
        dt = _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16
        order_p = None if query_order is None else _order_ptr(query_order, b * q)
        code = lib.gd4d_cross_attn_agg_items_fwd(
            ptrs, lv, cs, pyramid.pix_stride, pyramid.slice_stride, items_buf, _dev(agg, 'agg', f32),
            _dev(plan.wsum, 'wsum', f32), b, pyramid.rows // b, q, hh, 256, nl, plan.points, dt, order_p,
            int(slices[0]), int(slices[1]), _stream())
        _lib.check(code, 'gd4d_cross_attn_agg_items_fwd')
        return agg
    code = lib.gd4d_cross_attn_agg_sliced_fwd(
        ptrs, pyramid.slice_stride, _dev(plan.buf, 'plan', torch.uint8), _dev(agg, 'agg', f32), b, pyramid.rows // b, q, hh,
        256, nl, plan.points, _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16,
        None if query_order is None else _order_ptr(query_order, b * q), int(slices[0]), int(slices[1]), _stream())
    _lib.check(code, 'gd4d_cross_attn_agg_sliced_fwd')
    return agg


class CoarseValues:
    """Levels 2, 3 of a pyramid with one layer's value_proj already applied (bias included): `rows` (R, S23, 256) fp32,
    level 2's pixels first - what gd4d_value_proj_fwd over those two levels writes (GD4D_LAYOUT_PIXEL_MAJOR)."""

    def __init__(self, rows, level_hw):
        (h2, w2), (h3, w3) = level_hw
        if rows.dim() != 3 or rows.shape[1] != h2 * w2 + h3 * w3 or rows.shape[2] != 256 or rows.dtype != torch.float32 \
                or not rows.is_contiguous():
            raise ValueError(f'projected coarse levels {tuple(rows.shape)} inconsistent with {level_hw}')
        self.rows, self.level_hw = rows, [(int(h2), int(w2)), (int(h3), int(w3))]
        self.ptrs = [rows.data_ptr(), rows.data_ptr() + h2 * w2 * 1024]
        self.cam_stride = [rows.shape[1] * 1024] * 2


def coarse_supported(plan):
    """gd4d_cross_attn_agg_items_coarse_fwd: items plan, 4 levels, 8 heads."""
    return plan.items and len(plan.pyramid.level_hw) == 4 and plan.num_heads == 8


def cross_attn_agg_coarse_fwd(plan, coarse, agg=None, pagg=None):
    """gd4d_cross_attn_agg_items_coarse_fwd: levels 0, 1 gathered raw from the Plan's pyramid, levels 2, 3 from `coarse`
    (CoarseValues of THIS layer's value_proj).  Returns agg (B, Q, 8, 256), pagg (B, Q, 256); plan.wsum holds the fine levels'
    weight sums: value_proj_heads_fwd(agg, plan.wsum, W, b) + pagg is the layer's sampled value."""
    lib = _lib.load()
    pyramid = plan.pyramid
    dev = pyramid.device
    b, q, hh, query_order = plan.b, plan.q, plan.num_heads, plan.order
    if not coarse_supported(plan):
        raise _lib.Gd4dError('the coarse-projected gather takes an items plan over 4 levels with 8 heads')
    if list(coarse.level_hw) != [tuple(x) for x in pyramid.level_hw[2:]] or coarse.rows.shape[0] != pyramid.rows \
            or coarse.rows.device != dev:
        raise ValueError('projected coarse levels do not belong to this pyramid')
    f32 = torch.float32
    if agg is None:
        agg = torch.empty(b, q, hh, 256, device=dev, dtype=f32)
    if pagg is None:
        pagg = torch.empty(b, q, 256, device=dev, dtype=f32)
    ptrs = (ctypes.c_void_p * 4)(*pyramid.ptrs)
    lv = (ctypes.c_int32 * 8)(*[int(x) for hw in pyramid.level_hw for x in hw])
    cs = (ctypes.c_int64 * 4)(*pyramid.cam_stride)
    pp = (ctypes.c_void_p * 2)(*coarse.ptrs)
    pcs = (ctypes.c_int64 * 2)(*coarse.cam_stride)
    code = lib.gd4d_cross_attn_agg_items_coarse_fwd(
        ptrs, lv, cs, pyramid.pix_stride, pyramid.slice_stride, pp, pcs, _dev(plan.buf, 'plan', torch.uint8), _dev(agg, 'agg', f32),
        _dev(plan.wsum, 'wsum', f32), _dev(pagg, 'pagg', f32), b, pyramid.rows // b, q, hh, 256, 4, plan.points,
        _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16,
        None if query_order is None else _order_ptr(query_order, b * q), _stream())
    _lib.check(code, 'gd4d_cross_attn_agg_items_coarse_fwd')
    return agg, pagg


def value_proj_heads_fwd(agg, wsum, weight, bias=None, out=None):
    """gd4d_value_proj_heads_fwd: agg (..., Hh, 256), wsum (..., Hh) -> out (..., 256) = value_proj of the aggregates."""
    lib = _lib.load()
    hh, c = agg.shape[-2], agg.shape[-1]
    m = agg.numel() // (hh * c)
    f32 = torch.float32
    if out is None:
        out = torch.empty(*agg.shape[:-2], c, device=agg.device, dtype=f32)
    code = lib.gd4d_value_proj_heads_fwd(_dev(agg, 'agg', f32), _dev(wsum, 'wsum', f32), _dev(weight, 'weight', f32),
                                         _opt(bias, 'bias'), _dev(out, 'out', f32), m, hh, c, _stream())
    _lib.check(code, 'gd4d_value_proj_heads_fwd')
    return out


def value_proj_heads_bwd(grad_out, weight, bias=None, num_heads=8, grad_agg=None, beta=None):
    """gd4d_value_proj_heads_bwd: grad_out (..., 256) -> grad_agg (..., Hh, 256) = W_h^T grad_out[.., h], beta (..., Hh) =
    <b_h, grad_out[.., h]> (zeros without a bias): the gradient of value_proj_heads_fwd w.r.t. agg and wsum."""
    lib = _lib.load()
    c = grad_out.shape[-1]
    m = grad_out.numel() // c
    f32 = torch.float32
    if grad_agg is None:
        grad_agg = torch.empty(*grad_out.shape[:-1], num_heads, c, device=grad_out.device, dtype=f32)
    if beta is None:
        beta = torch.empty(*grad_out.shape[:-1], num_heads, device=grad_out.device, dtype=f32)
    code = lib.gd4d_value_proj_heads_bwd(_dev(grad_out, 'grad_out', f32), _dev(weight, 'weight', f32), _opt(bias, 'bias'),
                                         _dev(grad_agg, 'grad_agg', f32), _dev(beta, 'beta', f32), m, num_heads, c, _stream())
    _lib.check(code, 'gd4d_value_proj_heads_bwd')
    return grad_agg, beta


def value_proj_heads_bwd_weight(grad_out, agg, wsum=None, want_bias=True, into=None):
    """gd4d_value_proj_heads_bwd_weight: grad_out (..., 256), agg (..., Hh, 256), wsum (..., Hh) -> (grad_weight (256, 256),
    grad_bias (256) or None): value_proj's gradients from the per-head aggregates of the forward pass.  into=(w_buf, b_buf):
    added to these instead."""
    lib = _lib.load()
    f32 = torch.float32
    hh, c = agg.shape[-2], agg.shape[-1]
    m = grad_out.numel() // c
    if into is None:
        gw = torch.empty(c, c, device=grad_out.device, dtype=f32)
        gb = torch.empty(c, device=grad_out.device, dtype=f32) if want_bias else None
    else:
        gw, gb = into[0], (into[1] if want_bias else None)
    wsb = int(lib.gd4d_value_proj_heads_bwd_weight_workspace_bytes())
    ws = torch.empty(wsb, device=grad_out.device, dtype=torch.uint8)
    code = lib.gd4d_value_proj_heads_bwd_weight(_dev(grad_out, 'grad_out', f32), _dev(agg, 'agg', f32),
                                                _dev(wsum, 'wsum', f32) if want_bias else None, _dev(gw, 'grad_weight', f32),
                                                _dev(gb, 'grad_bias', f32) if want_bias else None, _dev(ws, 'workspace'),
                                                ctypes.c_size_t(wsb), m, hh, c, 0 if into is None else 1, _stream())
    _lib.check(code, 'gd4d_value_proj_heads_bwd_weight')
    return gw, gb


_VPH_WS = {}


def value_proj_heads_bwd_weight_group(problems, accumulate=True):
    """gd4d_value_proj_heads_bwd_weight_group: problems = list (<= 8) of (grad_out (..., 256), agg (..., Hh, 256), wsum (..., Hh),
    grad_weight (256, 256), grad_bias (256) or None) - value_proj's gradients of several layers in one pair of launches, added to
    (accumulate) or written into the targets."""
    lib = _lib.load()
    f32 = torch.float32
    n = len(problems)
    hh, c = problems[0][1].shape[-2], problems[0][1].shape[-1]
    dev = problems[0][0].device
    nbytes = n * int(lib.gd4d_value_proj_heads_bwd_weight_workspace_bytes())
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _VPH_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _VPH_WS[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    arr = lambda v: (ctypes.c_void_p * n)(*v)          # noqa: E731
    rows = (ctypes.c_int32 * n)(*[int(g.numel() // c) for g, _, _, _, _ in problems])
    code = lib.gd4d_value_proj_heads_bwd_weight_group(
        arr([_dev(g, 'grad_out', f32).value for g, _, _, _, _ in problems]), arr([_dev(a, 'agg', f32).value for _, a, _, _, _ in problems]),
        arr([_dev(w, 'wsum', f32).value for _, _, w, _, _ in problems]), arr([_dev(gw, 'grad_weight', f32).value for _, _, _, gw, _ in problems]),
        arr([None if gb is None else _dev(gb, 'grad_bias', f32).value for _, _, _, _, gb in problems]), rows, n,
        _dev(ws, 'workspace'), ctypes.c_size_t(nbytes), hh, c, 1 if accumulate else 0, _stream())
    _lib.check(code, 'gd4d_value_proj_heads_bwd_weight_group')


def cross_attn_dot_bytes(b, n, q, num_heads, points=4):
    return int(_lib.load().gd4d_cross_attn_dot_bytes(b, n, q, num_heads, points))


def _wgrad_arrays(problems):
    """The C arrays of gd4d_linear_bwd_weight_group for problems = [(x (M, K), grad_y (M, N), grad_w (N, K), grad_b (N) or None)]."""
    f32 = torch.float32
    n = len(problems)
    xs, gys, gws, gbs, dims = [], [], [], [], []
    for x, gy, gw, gb in problems:
        k, nn_ = x.shape[-1], gy.shape[-1]
        m = x.numel() // k
        if gy.numel() // nn_ != m or tuple(gw.shape) != (nn_, k) or (gb is not None and gb.numel() != nn_):
            raise ValueError('weight-gradient group: inconsistent shapes')
        xs.append(_dev(x, 'x', f32).value); gys.append(_dev(gy, 'grad_y', f32).value); gws.append(_dev(gw, 'grad_w', f32).value)
        gbs.append(_dev(gb, 'grad_b', f32).value if gb is not None else None)
        dims += [m, k, nn_, k, nn_]
    arr = lambda v: (ctypes.c_void_p * n)(*v)          # noqa: E731
    return arr(xs), arr(gys), arr(gws), arr(gbs), (ctypes.c_int32 * (5 * n))(*dims), n


def cross_attn_dot_sliced(plan, grad_agg, dpart=None, wgrads=None):
    """gd4d_cross_attn_dot_sliced: D[pair] = <grad_agg[q, h], raw pixel of the pair> for every pair of `plan`, as 8 per-slice
    partials (uint8 buffer of gd4d_cross_attn_dot_bytes; only the passes the plan uses are written).
    wgrads: up to 16 weight-gradient problems (as linear_bwd_weight_group takes them; added to their targets) whose tiles ride
    in the launch (gd4d_cross_attn_dot_sliced_wgrad; 8 heads, 4 levels, fp32: wgrads_ride_with(plan) says whether)."""
    lib = _lib.load()
    plan.need_pairs('gd4d_cross_attn_dot_sliced')
    pyramid = plan.pyramid
    b, q, hh = plan.b, plan.q, plan.num_heads
    n = pyramid.rows // b
    nl = len(pyramid.level_hw)
    nbytes = int(lib.gd4d_cross_attn_dot_bytes(b, n, q, hh, plan.points))
    if dpart is None:
        dpart = torch.empty(nbytes, device=pyramid.device, dtype=torch.uint8)
    ptrs = (ctypes.c_void_p * nl)(*pyramid.ptrs)
    if wgrads:
        xs, gys, gws, gbs, dims, cnt = _wgrad_arrays(wgrads)
        code = lib.gd4d_cross_attn_dot_sliced_wgrad(
            ptrs, pyramid.slice_stride, _dev(plan.buf, 'plan', torch.uint8), _dev(grad_agg, 'grad_agg', torch.float32),
            _dev(dpart, 'dpart', torch.uint8), dpart.numel(), b, n, q, hh, 256, nl, plan.points,
            _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16,
            None if plan.order is None else _order_ptr(plan.order, b * q), xs, gys, gws, gbs, dims, cnt, 1, _stream())
        _lib.check(code, 'gd4d_cross_attn_dot_sliced_wgrad')
        return dpart
    code = lib.gd4d_cross_attn_dot_sliced(
        ptrs, pyramid.slice_stride, _dev(plan.buf, 'plan', torch.uint8), _dev(grad_agg, 'grad_agg', torch.float32),
        _dev(dpart, 'dpart', torch.uint8), dpart.numel(), b, n, q, hh, 256, nl, plan.points,
        _lib.F32 if pyramid.dtype == torch.float32 else _lib.BF16,
        None if plan.order is None else _order_ptr(plan.order, b * q), _stream())
    _lib.check(code, 'gd4d_cross_attn_dot_sliced')
    return dpart


def wgrads_ride_with(plan):
    """Whether cross_attn_dot_sliced(plan, ..., wgrads=...) has a launch for this plan's shape."""
    return plan.num_heads == 8 and len(plan.pyramid.level_hw) == 4 and plan.pyramid.dtype == torch.float32


def cross_attn_plan_bwd(plan, dpart, beta, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w,
                        raw_cam_weights=False, status=None):
    """gd4d_cross_attn_plan_bwd: the query-side gradients of the sliced path from D (cross_attn_dot_sliced) and beta.
    Returns (grad_ref, grad_offsets, grad_attn_logits, grad_cam_logits) - what cross_attn_bwd returns after grad_value."""
    lib = _lib.load()
    plan.need_pairs('gd4d_cross_attn_plan_bwd')
    f32 = torch.float32
    b, q, hh = plan.b, plan.q, plan.num_heads
    n = lidar2img.shape[1]
    p = offsets.shape[3]
    level_hw = plan.pyramid.level_hw
    nl = len(level_hw)
    dev = ref.device
    gr = torch.empty(b, q, 3, device=dev, dtype=f32)
    go = torch.empty(b, q, hh, p, 3, device=dev, dtype=f32)
    ga = torch.empty(b, q, hh, nl, p, device=dev, dtype=f32)
    gc = torch.empty(b, q, n, device=dev, dtype=f32)
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    nbytes = lib.gd4d_cross_attn_bwd_workspace_bytes(b, q, hh, nl, p)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8) if nbytes else None
    code = lib.gd4d_cross_attn_plan_bwd(
        _dev(ref, 'ref', f32), _dev(offsets, 'offsets', f32), _dev(attn_logits, 'attn_logits', f32),
        _dev(cam_logits, 'cam_logits', f32), _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w), lv,
        _dev(plan.buf, 'plan', torch.uint8), _dev(dpart, 'dpart', torch.uint8), _opt(beta, 'beta'), _dev(gr, 'grad_ref'),
        _dev(go, 'grad_offsets'), _dev(ga, 'grad_attn_logits'), _dev(gc, 'grad_cam_logits'),
        None if ws is None else _dev(ws, 'workspace'), ctypes.c_size_t(nbytes),
        None if status is None else _dev(status, 'status', torch.int32), b, n, q, hh, nl, p, 1 if raw_cam_weights else 0,
        None if plan.order is None else _order_ptr(plan.order, b * q), _stream())
    _lib.check(code, 'gd4d_cross_attn_plan_bwd')
    return gr, go, ga, gc


_CHUNK_WALKS = {}


def _chunk_walk(level_hw, rows, device, regions=(6, 10)):
    """The order gd4d_pyramid_grad_reduce walks the chunks in: camera by camera, and inside a camera image region by image
    region (regions[0] x regions[1] per image) with ALL levels of a region together - the workgroups an XCD runs at one
    time then need the table rows of the few queries that look at one region, not of a whole camera (level-major, the
    64 chunks in flight per XCD cover most of a camera's coarse maps: 77 % L2 hits, 2.8 GB from the fabric per pass).
    Cached per geometry (an int32 permutation on the device)."""
    import numpy as np
    key = (tuple(level_hw), int(rows), str(device), regions)
    hit = _CHUNK_WALKS.get(key)
    if hit is not None:
        return hit
    lib = _lib.load()
    nl = len(level_hw)
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    geo = (ctypes.c_int32 * (5 * nl))()
    _lib.check(lib.gd4d_pyramid_grad_chunk_geometry(lv, int(rows), nl, geo), 'gd4d_pyramid_grad_chunk_geometry')
    keys, ids = [], []
    span = max(int(geo[5 * l + 2]) * int(geo[5 * l + 3]) for l in range(nl))
    for l, (h, w) in enumerate(level_hw):
        cws, chs, cw_n, ch_n, base = geo[5 * l:5 * l + 5]
        r, cy, cx = np.meshgrid(np.arange(rows), np.arange(ch_n), np.arange(cw_n), indexing='ij')
        v = np.minimum((cy + 0.5) * (1 << chs) / h, 0.999999)
        u = np.minimum((cx + 0.5) * (1 << cws) / w, 0.999999)
        ry, rx = (v * regions[0]).astype(np.int64), (u * regions[1]).astype(np.int64)
        rx = np.where(ry % 2 == 1, regions[1] - 1 - rx, rx)                        # boustrophedon: neighbours stay neighbours
        k = (((r * regions[0] + ry) * regions[1] + rx) * nl + (nl - 1 - l)) * span + cy * cw_n + cx
        keys.append(k.reshape(-1))
        ids.append((base + (r * ch_n + cy) * cw_n + cx).reshape(-1))
    keys, ids = np.concatenate(keys), np.concatenate(ids)
    walk = torch.from_numpy(ids[np.argsort(keys, kind='stable')].astype(np.int32)).to(device)
    _CHUNK_WALKS[key] = walk
    return walk


class PyramidGrad:
    """The gradient of an NCHW pyramid from the plans of all decoder layers (gd4d_pyramid_grad_count / _scan / _fill /
    _reduce): add_layer() per layer (in any order, each with its plan and its grad_agg rows), finish() once.

    Buffers are sized for `layers` layers of B*Q*Hh rows (alloc_table with a list: per-layer Q); the slot / record buffers by
    the plans' capacity (8 bytes per pair a plan can hold - only what the counts say is touched)."""

    def __init__(self, pyramid, layers, b, q, num_heads, chunk_walk=True, points=4):
        self.pyramid, self.layers, self.b, self.q, self.hh = pyramid, int(layers), int(b), int(q), int(num_heads)
        self.points = int(points)                 # points per head of the plans this sink takes (4, or 8 with 8 heads)
        dev = pyramid.device
        lib = _lib.load()
        self.n = pyramid.rows // self.b
        nl = len(pyramid.level_hw)
        self._lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in pyramid.level_hw for x in hw])
        self._cs = (ctypes.c_int64 * nl)(*pyramid.cam_stride)
        self.chunks = int(lib.gd4d_pyramid_grad_chunks(self._lv, pyramid.rows, nl))
        if self.chunks <= 0:
            raise _lib.Gd4dError('gd4d_pyramid_grad_chunks: unsupported pyramid')
        self.count = torch.zeros(self.chunks, device=dev, dtype=torch.int32)
        self._scanned, self._riding = None, []
        self.table, self.layer_q = None, {}
        if self.layers > 0:
            self.alloc_table(self.layers)
        self.slot_bytes = int(lib.gd4d_pyramid_grad_slots_bytes(self.b, self.n, self.q, self.hh, self.points))
        self.plans, self.prepared = [], None
        self.order = _chunk_walk(pyramid.level_hw, pyramid.rows, dev) if chunk_walk else None

    def alloc_table(self, layers):
        """The grad_agg table (zeros: a layer whose backward never runs contributes nothing).  layers: a count (every layer
        has the constructor's Q) or a list of per-layer query counts - the passes of Detr3DTransformer.forward_shared share
        one pyramid but need not have the same number of queries (teacher_queries, detr3d_head_pe.py:560-566)."""
        qs = [self.q] * int(layers) if isinstance(layers, int) else [int(x) for x in layers]
        for layer, q in self.layer_q.items():
            if layer < len(qs) and qs[layer] != q:
                raise _lib.Gd4dError(f'PyramidGrad: layer {layer} was counted with {q} queries, the table is asked for {qs[layer]}')
        self.layers = len(qs)
        self.table_q = qs
        self.row_base = [0]
        for q in qs:
            self.row_base.append(self.row_base[-1] + self.b * q * self.hh)
        if self.row_base[-1] >= 1 << 26:
            raise _lib.Gd4dError('PyramidGrad: more than 2^26 table rows (a record keeps its row in 26 bits)')
        self.table = torch.zeros(max(self.row_base[-1], 1), 256, device=self.pyramid.device, dtype=torch.float32)

    def grad_agg_rows(self, layer):
        """(B, Q_layer, Hh, 256) view of the table: where layer `layer`'s gd4d_value_proj_heads_bwd writes."""
        return self.table[self.row_base[layer]:self.row_base[layer + 1]].view(self.b, self.table_q[layer], self.hh, 256)

    def add_layer(self, layer, plan):
        """Hand every record of `plan` its slot (the plan is kept until finish(): its buffer must not be overwritten).  The
        plan's own (B, Q, Hh) say where its pairs are; B and Hh must be the sink's (they fix the pyramid rows / the table's
        row width), Q is per layer."""
        lib = _lib.load()
        slots, slot_bytes = self.begin_layer(layer, plan)
        layer = int(layer)
        code = lib.gd4d_pyramid_grad_count(_dev(plan.buf, 'plan', torch.uint8), self._lv, self._cs, self.pyramid.pix_stride,
                                           _dev(self.count, 'count', torch.int32), _dev(slots, 'slots'), ctypes.c_size_t(slot_bytes),
                                           self.b, self.n, plan.q, self.hh, len(self.pyramid.level_hw), plan.points, _stream())
        _lib.check(code, 'gd4d_pyramid_grad_count')
        self.plans.append((layer, plan, slots))

    def begin_layer(self, layer, plan):
        """The checks of add_layer and the layer's slot buffer (the count itself: add_layer, or the forward gather's launch -
        cross_attn_agg_sliced_fwd(count=...), which then appends to self.plans)."""
        lib = _lib.load()
        plan.need_pairs('gd4d_pyramid_grad_count')
        layer = int(layer)
        if plan.b != self.b or plan.num_heads != self.hh or plan.pyramid.rows != self.pyramid.rows:
            raise _lib.Gd4dError(f'PyramidGrad: plan of (B, Hh, rows) = ({plan.b}, {plan.num_heads}, {plan.pyramid.rows}) handed to a '
                                 f'sink of ({self.b}, {self.hh}, {self.pyramid.rows})')
        if self.table is not None and (layer >= self.layers or self.table_q[layer] != plan.q):
            raise _lib.Gd4dError(f'PyramidGrad: layer {layer} has {plan.q} queries, the table was made for '
                                 f'{self.table_q[layer] if layer < self.layers else "fewer layers"}')
        self.layer_q[layer] = plan.q
        lib = _lib.load()
        if plan.points != self.points:
            raise _lib.Gd4dError(f'PyramidGrad: a plan with {plan.points} points per head handed to a sink made for {self.points}')
        slot_bytes = int(lib.gd4d_pyramid_grad_slots_bytes(self.b, self.n, plan.q, self.hh, plan.points))
        slots = torch.empty(slot_bytes, device=self.pyramid.device, dtype=torch.uint8)
        return slots, slot_bytes

    def prepare(self):
        """scan + fill + sort: the layers' records bucketed by chunk and grouped by pixel.  Needs the counts of every layer
        (add_layer) and nothing from the backward pass."""
        if self._scanned is None:
            self.scan()
        self.finish_prepare()

    def scan(self):
        """The first part of prepare(): the chunks' starts.  The fills may then ride in other launches (take_fills ->
        mha_core_bwd(fills=...)); finish_prepare() launches whatever is left of them, and the sort."""
        lib = _lib.load()
        dev = self.pyramid.device
        i32 = torch.int32
        start = torch.empty(self.chunks, device=dev, dtype=i32)
        wsb = int(lib.gd4d_pyramid_grad_scan_workspace_bytes(self.chunks))
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        code = lib.gd4d_pyramid_grad_scan(_dev(self.count, 'count', i32), _dev(start, 'start', i32), _dev(ws, 'workspace'),
                                          ctypes.c_size_t(wsb), self.chunks, _stream())
        _lib.check(code, 'gd4d_pyramid_grad_scan')
        if self.table is None:
            raise _lib.Gd4dError('PyramidGrad.prepare: alloc_table() first (the records carry table rows)')
        nbytes = max(sum(slots.numel() for _, _, slots in self.plans), self.slot_bytes)
        self._scanned = (start, torch.empty(nbytes, device=dev, dtype=torch.uint8), nbytes)
        for layer, plan, slots in self.plans:
            if layer >= self.layers or self.table_q[layer] != plan.q:
                raise _lib.Gd4dError(f'PyramidGrad: layer {layer} ({plan.q} queries) does not match the table')

    def take_fills(self, n):
        """Up to n (<= 2) pending fills for another launch to carry: ([(plan, slots, first table row)], start, records, B, N, Hh),
        or None when none is left."""
        n = min(int(n), 2, len(self.plans))
        if n <= 0 or self._scanned is None:
            return None
        start, records, _ = self._scanned
        jobs = [(plan, slots, self.row_base[layer]) for layer, plan, slots in self.plans[:n]]
        self._riding += self.plans[:n]                     # (their buffers live until finish_prepare)
        self.plans = self.plans[n:]
        return jobs, start, records, self.b, self.n, self.hh

    def finish_prepare(self):
        lib = _lib.load()
        dev = self.pyramid.device
        i32 = torch.int32
        start, records, nbytes = self._scanned
        for layer, plan, slots in self.plans:
            code = lib.gd4d_pyramid_grad_fill(
                _dev(plan.buf, 'plan', torch.uint8), _dev(slots, 'slots'), _dev(start, 'start', i32), _dev(records, 'records'),
                self.row_base[layer], None if plan.order is None else _order_ptr(plan.order, self.b * plan.q),
                self.b, self.n, plan.q, self.hh, plan.points, _stream())
            _lib.check(code, 'gd4d_pyramid_grad_fill')
        sorted_ = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        pxoff = torch.empty(self.chunks, 65, device=dev, dtype=i32)
        code = lib.gd4d_pyramid_grad_sort(_dev(self.count, 'count', i32), _dev(start, 'start', i32), _dev(records, 'records'),
                                          _dev(sorted_, 'sorted'), _dev(pxoff, 'pxoff', i32), self.chunks, _stream())
        _lib.check(code, 'gd4d_pyramid_grad_sort')
        self.prepared = (start, pxoff, sorted_)
        self.plans = []
        self._riding = []
        self._scanned = None

    def reduce(self, grads=None, channels_last=False):
        """-> L tensors (R, 256, H_l, W_l) fp32: the pyramid's gradient summed over the layers (prepare() first; the table
        rows of every layer must have been written).  channels_last: the tensors are stored (R, H_l, W_l, 256) - the layout of
        levels the gather read in place - and returned as (R, 256, H_l, W_l) views of that memory."""
        lib = _lib.load()
        py = self.pyramid
        nl = len(py.level_hw)
        i32 = torch.int32
        start, pxoff, sorted_ = self.prepared
        if grads is None:
            grads = [torch.empty((py.rows, h, w, 256) if channels_last else (py.rows, 256, h, w), device=py.device, dtype=torch.float32)
                     for h, w in py.level_hw]
        for g in grads:
            _dev(g, 'grads', torch.float32)
        ptrs = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grads])
        code = lib.gd4d_pyramid_grad_reduce(_dev(start, 'start', i32), _dev(pxoff, 'pxoff', i32), _dev(sorted_, 'sorted'),
                                            _dev(self.table, 'table', torch.float32), ptrs, self._lv,
                                            None if self.order is None else _dev(self.order, 'chunk_order', i32), py.rows, 256, nl,
                                            1 if channels_last else 0, _stream())
        _lib.check(code, 'gd4d_pyramid_grad_reduce')
        self.prepared = None
        return [g.permute(0, 3, 1, 2) for g in grads] if channels_last else grads

    def finish(self, grads=None, channels_last=False):
        self.prepare()
        return self.reduce(grads, channels_last)


def _order_ptr(order, count):
    if order.dtype != torch.int32 or order.numel() != count:
        raise ValueError(f'query_order must be an int32 permutation of {count} entries')
    return _dev(order, 'query_order')


def query_order_fwd(ref, pc_range, out=None):
    """gd4d_query_order_fwd: ref (B,Q,3) in [0,1] -> int32 (B*Q) locality order for cross_attn_fwd(query_order=)."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    if out is None:
        out = torch.empty(b * q, device=ref.device, dtype=torch.int32)
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    code = lib.gd4d_query_order_fwd(_dev(ref, 'ref', torch.float32), rng, _order_ptr(out, b * q), b, q, _stream())
    _lib.check(code, 'gd4d_query_order_fwd')
    return out


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


def detr3d_bwd(feats, ref, attn_logits, lidar2img, pc_range, img_h, img_w, grad_out, want_feats=True, want_ref=True):
    """gd4d_detr3d_bwd: gradients of detr3d_fwd(...)['out'].  Returns (grad_feats: list like feats or None,
    grad_logits like attn_logits, grad_ref (B, Q, 3) or None)."""
    lib = _lib.load()
    b, n, c = feats[0].shape[:3]
    q = ref.shape[1]
    nl = len(feats)
    f32 = torch.float32
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    gf = [torch.zeros_like(f) for f in feats] if want_feats else None
    gptrs = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gf]) if want_feats else None
    gl = torch.empty(b * q * n * nl, device=ref.device, dtype=f32).view_as(attn_logits)
    gr = torch.empty(b, q, 3, device=ref.device, dtype=f32) if want_ref else None
    code = lib.gd4d_detr3d_bwd(ptrs, lv, _dev(ref, 'ref', f32), _dev(attn_logits, 'attn_logits', f32),
                               _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w), _dev(grad_out, 'grad_out', f32),
                               gptrs, _dev(gl, 'grad_logits'), _dev(gr, 'grad_ref') if want_ref else None, b, n, q, c, nl, 1, _stream())
    _lib.check(code, 'gd4d_detr3d_bwd')
    return gf, gl, gr


def detr3d_v2_fwd(feats, ref, attn_logits, offsets, lidar2img, pc_range, img_h, img_w, num_heads, want_mask=False):
    """gd4d_detr3d_v2_fwd.  feats: list of L tensors (B, N, C, H_l, W_l) fp32; attn_logits (B, Q, N, Hh, L*P);
    offsets (B, Q, N, Hh, L, P, 2), P == L.  Returns out (B, Q, C) [, mask (B, N, Q) uint8]."""
    lib = _lib.load()
    b, n, c = feats[0].shape[:3]
    q = ref.shape[1]
    nl = len(feats)
    if attn_logits.numel() != b * q * n * num_heads * nl * nl or offsets.numel() != 2 * attn_logits.numel():
        raise ValueError('attn_logits / offsets must be (B, Q, N, heads, L*P) / (..., L, P, 2) with P == L')
    f32 = torch.float32
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    out = torch.empty(b, q, c, device=ref.device, dtype=f32)
    mask = torch.empty(b, n, q, device=ref.device, dtype=torch.uint8) if want_mask else None
    code = lib.gd4d_detr3d_v2_fwd(ptrs, lv, _dev(ref, 'ref', f32), _dev(attn_logits, 'attn_logits', f32),
                                  _dev(offsets, 'offsets', f32), _dev(lidar2img, 'lidar2img', f32), rng, float(img_h),
                                  float(img_w), _dev(out, 'out'), _dev(mask, 'mask') if want_mask else None,
                                  b, n, q, c, int(num_heads), nl, nl, _stream())
    _lib.check(code, 'gd4d_detr3d_v2_fwd')
    return (out, mask) if want_mask else out


def detr3d_v2_bwd(feats, ref, attn_logits, offsets, lidar2img, pc_range, img_h, img_w, num_heads, grad_out, want_feats=True,
                  want_ref=True):
    """gd4d_detr3d_v2_bwd: the gradients of detr3d_v2_fwd's `out`.  Returns (grad_feats list or None, grad_attn_logits
    (B, Q, N, Hh, L*P), grad_offsets (B, Q, N, Hh, L, P, 2), grad_ref (B, Q, 3) or None).  C <= 256."""
    lib = _lib.load()
    b, n, c = feats[0].shape[:3]
    q = ref.shape[1]
    nl = len(feats)
    f32 = torch.float32
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    gfeats = [torch.zeros_like(f) for f in feats] if want_feats else None
    gptrs = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gfeats]) if want_feats else None
    gl = torch.empty(b, q, n, num_heads, nl * nl, device=ref.device, dtype=f32)
    go = torch.empty(b, q, n, num_heads, nl, nl, 2, device=ref.device, dtype=f32)
    gr = torch.empty(b, q, 3, device=ref.device, dtype=f32) if want_ref else None
    code = lib.gd4d_detr3d_v2_bwd(ptrs, lv, _dev(ref, 'ref', f32), _dev(attn_logits, 'attn_logits', f32),
                                  _dev(offsets, 'offsets', f32), _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w),
                                  _dev(grad_out, 'grad_out', f32), gptrs, _dev(gl, 'grad_logits'), _dev(go, 'grad_offsets'),
                                  _dev(gr, 'grad_ref') if want_ref else None, b, n, q, c, nl, int(num_heads), _stream())
    _lib.check(code, 'gd4d_detr3d_v2_bwd')
    return gfeats, gl, go, gr


def _vp_workspace(nlayers, device):
    """Scratch for the weight fragments of a value_proj launch (rewritten by every call: stream-ordered, so a fresh
    tensor per call keeps concurrent launches on different streams apart; the caching allocator makes it free)."""
    nbytes = _lib.load().gd4d_value_proj_workspace_bytes(int(nlayers))
    return torch.empty(nbytes, device=device, dtype=torch.uint8), nbytes


def value_proj_fwd(feats, weight, bias, out_dtype=torch.float32, out=None, num_heads=8, head_major=False,
                   bf16_math=False, max_cus=0):
    """gd4d_value_proj_fwd.  feats: list of L tensors (B, N, C, H_l, W_l) or (R, C, H_l, W_l) fp32;
    weight (C, C); bias (C) or None.  Returns (R, S, C) in `out_dtype`, or (R, Hh, S, C/Hh) with
    head_major=True.  max_cus: occupy at most that many CUs (0 = all)."""
    o = value_proj_multi_fwd(feats, [weight], [bias], out_dtype, num_heads, head_major, bf16_math, max_cus,
                             outs=None if out is None else [out])
    return o[0]


def value_proj_multi_fwd(feats, weights, biases, out_dtype=torch.float32, num_heads=8, head_major=False,
                         bf16_math=False, max_cus=0, outs=None):
    """gd4d_value_proj_multi_fwd: project the same pyramid with NL (weight, bias) pairs in one
    launch.  Returns a list of NL tensors (R, S, C) (or (R, Hh, S, C/Hh) with head_major=True)."""
    lib = _lib.load()
    f32 = torch.float32
    nlayers = len(weights)
    c = weights[0].shape[0]
    r = feats[0].numel() // (c * feats[0].shape[-1] * feats[0].shape[-2])
    nl = len(feats)
    s = sum(f.shape[-1] * f.shape[-2] for f in feats)
    shape = (r, num_heads, s, c // num_heads) if head_major else (r, s, c)
    if outs is None:
        outs = [torch.empty(*shape, device=weights[0].device, dtype=out_dtype) for _ in range(nlayers)]
    ws, nbytes = _vp_workspace(nlayers, weights[0].device)
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    wp = (ctypes.c_void_p * nlayers)(*[_dev(w, 'weight', f32).value for w in weights])
    bp = (ctypes.c_void_p * nlayers)(*[(_dev(b, 'bias', f32).value if b is not None else None)
                                       for b in biases])
    op = (ctypes.c_void_p * nlayers)(*[_dev(o, 'out').value for o in outs])
    code = lib.gd4d_value_proj_multi_fwd(ptrs, lv, wp, bp, op, r, c, nl, nlayers, num_heads, _lib.F32,
                                         _value_dtype(outs[0]),
                                         _lib.HEAD_MAJOR if head_major else _lib.PIXEL_MAJOR,
                                         int(bool(bf16_math)), _dev(ws, 'workspace'), ctypes.c_size_t(nbytes),
                                         int(max_cus), _stream())
    _lib.check(code, 'gd4d_value_proj_multi_fwd')
    return outs


def linear_bwd_weight(x, grad_y, want_bias=True, into=None):
    """gd4d_linear_bwd_weight.  x (..., K), grad_y (..., N) fp32 with the same leading shape (contiguous rows).
    Returns (grad_w (N, K), grad_b (N) or None).  into=(w_buf, b_buf or None): the sums are ADDED to these tensors (views
    of a flat gradient buffer) and returned."""
    lib = _lib.load()
    f32 = torch.float32
    k, n = x.shape[-1], grad_y.shape[-1]
    m = x.numel() // k
    if grad_y.numel() // n != m:
        raise ValueError('x and grad_y must have the same number of rows')
    if into is None:
        gw = torch.empty(n, k, device=x.device, dtype=f32)
        gb = torch.empty(n, device=x.device, dtype=f32) if want_bias else None
    else:
        gw, gb = into
        if tuple(gw.shape) != (n, k) or (want_bias and (gb is None or gb.numel() != n)):
            raise ValueError('linear_bwd_weight: the accumulation targets do not match the gradient shapes')
        gb = gb if want_bias else None
    code = lib.gd4d_linear_bwd_weight(_dev(x, 'x', f32), _dev(grad_y, 'grad_y', f32), _dev(gw, 'grad_w', f32),
                                      _dev(gb, 'grad_b', f32) if gb is not None else None, m, k, n, k, n, 0 if into is None else 1,
                                      _stream())
    _lib.check(code, 'gd4d_linear_bwd_weight')
    return gw, gb


def linear_bwd_weight_group(problems, accumulate=True):
    """gd4d_linear_bwd_weight_group: problems = list (<= 16) of (x (M, K), grad_y (M, N), grad_w (N, K), grad_b (N) or None);
    the sums are added to (accumulate) or written into grad_w / grad_b.  One launch for all of them."""
    lib = _lib.load()
    xs, gys, gws, gbs, dims, n = _wgrad_arrays(problems)
    code = lib.gd4d_linear_bwd_weight_group(xs, gys, gws, gbs, dims, n, 1 if accumulate else 0, _stream())
    _lib.check(code, 'gd4d_linear_bwd_weight_group')


_VP_BWD_WS = {}


def value_proj_bwd_input(grad_out, weight, shapes, grads=None, accumulate=False):
    """gd4d_value_proj_bwd_input.  grad_out (R, S, C) fp32; weight (C, C); shapes: per level (H_l, W_l).
    Returns the list of L gradients (R, C, H_l, W_l); with `grads` given they are written (accumulate=False) or added
    to (accumulate=True) in place."""
    lib = _lib.load()
    f32 = torch.float32
    c = weight.shape[0]
    r = grad_out.numel() // (c * sum(h * w for h, w in shapes))
    nl = len(shapes)
    if grads is None:
        if accumulate:
            raise ValueError('accumulate=True needs the tensors to add to')
        grads = [torch.empty(r, c, h, w, device=grad_out.device, dtype=f32) for h, w in shapes]
    ptrs = (ctypes.c_void_p * nl)(*[_dev(g, f'grads[{i}]', f32).value for i, g in enumerate(grads)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in shapes for x in hw])
    code = lib.gd4d_value_proj_bwd_input(_dev(grad_out, 'grad_out', f32), _dev(weight, 'weight', f32), ptrs, lv,
                                         r, c, nl, int(bool(accumulate)), _stream())
    _lib.check(code, 'gd4d_value_proj_bwd_input')
    return grads


def value_proj_bwd_weight(grad_out, feats, want_bias=True):
    """gd4d_value_proj_bwd_weight.  grad_out (R, S, C) fp32; feats: the L NCHW levels the forward read.
    Returns (grad_weight (C, C), grad_bias (C) or None)."""
    lib = _lib.load()
    f32 = torch.float32
    c = feats[0].shape[-3]
    nl = len(feats)
    r = feats[0].numel() // (c * feats[0].shape[-1] * feats[0].shape[-2])
    dev = grad_out.device
    nbytes = lib.gd4d_value_proj_bwd_weight_workspace_bytes()
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _VP_BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _VP_BWD_WS[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    gw = torch.empty(c, c, device=dev, dtype=f32)
    gb = torch.empty(c, device=dev, dtype=f32) if want_bias else None
    ptrs = (ctypes.c_void_p * nl)(*[_dev(f, f'feats[{i}]', f32).value for i, f in enumerate(feats)])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[-2:]])
    code = lib.gd4d_value_proj_bwd_weight(_dev(grad_out, 'grad_out', f32), ptrs, lv, _dev(gw, 'grad_weight'),
                                          _dev(gb, 'grad_bias') if want_bias else None, _dev(ws, 'workspace'),
                                          ctypes.c_size_t(nbytes), r, c, nl, _stream())
    _lib.check(code, 'gd4d_value_proj_bwd_weight')
    return gw, gb


def _opt(t, name):
    return _dev(t, name, torch.float32) if t is not None else None


def linear_fwd(x, weight, bias=None, x2=None, n_split=None, relu=False, r1=None, r2=None, out=None,
               inv_sigmoid_in=False, weight_kn=False, want_xsum=False):
    """gd4d_linear_fwd on the last dimension: y = act((x [+ x2 for cols < n_split]) W^T + b) [+r1] [+r2].
    x (..., K) contiguous - or 2-D with a row stride (a column slice of a wider buffer, e.g. the q|k part of a packed
    gradient); weight (N, K); residuals (..., N) contiguous.  Returns (..., N); with want_xsum (needs x2): (y, x + x2)."""
    lib = _lib.load()
    k = x.shape[-1]
    n = weight.shape[1] if weight_kn else weight.shape[0]      # weight_kn: weight is (K, N) - y = x W (a Linear's dgrad)
    if weight_kn and weight.shape[0] != k:
        raise ValueError('weight_kn: weight must be (K, N) with K = x.shape[-1]')
    m = x.numel() // k
    ldx = k
    if not x.is_contiguous():
        if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) < k or x2 is not None:
            raise ValueError('x must be contiguous, or 2-D with unit column stride (and no x2)')
        ldx = x.stride(0)
    if out is None:
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    if x2 is not None and x2.shape != x.shape:
        raise ValueError('x2 must have the shape of x')
    xsum = None
    if want_xsum:
        if x2 is None:
            raise ValueError('want_xsum needs x2')
        xsum = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    xptr = _dev(x, 'x', torch.float32) if ldx == k else _devptr_strided(x, 'x')
    code = lib.gd4d_linear_fwd(xptr, _opt(x2, 'x2'), _dev(weight, 'weight', torch.float32),
                               _opt(bias, 'bias'), _opt(r1, 'r1'), _opt(r2, 'r2'), _dev(out, 'out'),
                               m, k, n, n if n_split is None else int(n_split),
                               int(bool(relu)) | (2 if inv_sigmoid_in else 0) | (8 if weight_kn else 0),
                               ldx, n, n, n, None if xsum is None else _dev(xsum, 'xsum'), _stream())
    _lib.check(code, 'gd4d_linear_fwd')
    return (out, xsum) if want_xsum else out


def _devptr_strided(t, name):
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.Gd4dError(f'{name}: fp32 device tensor expected, got {t.dtype} on {t.device}')
    return ctypes.c_void_p(t.data_ptr())


def linear_group_fwd(x, weights, biases, x2=None, want_xsum=False):
    """gd4d_linear_group_fwd: [(x + x2) W_g^T + b_g for g] in one launch.  x (..., K) contiguous; returns a list
    (want_xsum, needs x2: (list, x + x2))."""
    lib = _lib.load()
    g = len(weights)
    k = x.shape[-1]
    m = x.numel() // k
    if x2 is not None and x2.shape != x.shape:
        raise ValueError('x2 must have the shape of x')
    if want_xsum and x2 is None:
        raise ValueError('want_xsum needs x2')
    outs = [torch.empty(*x.shape[:-1], w.shape[0], device=x.device, dtype=torch.float32) for w in weights]
    xsum = torch.empty(x.shape, device=x.device, dtype=torch.float32) if want_xsum else None
    vp = ctypes.c_void_p
    warr = (vp * g)(*[_dev(w, 'weight', torch.float32).value for w in weights])
    barr = (vp * g)(*[None if b is None else _dev(b, 'bias', torch.float32).value for b in biases])
    yarr = (vp * g)(*[o.data_ptr() for o in outs])
    narr = (ctypes.c_int32 * g)(*[w.shape[0] for w in weights])
    code = lib.gd4d_linear_group_fwd(_dev(x, 'x', torch.float32), _opt(x2, 'x2'), warr, barr, yarr, narr, g, m, k, k,
                                     None if xsum is None else _dev(xsum, 'xsum'), _stream())
    _lib.check(code, 'gd4d_linear_group_fwd')
    return (outs, xsum) if want_xsum else outs


def layernorm_fwd(x, gamma, beta, eps=1e-5, res=None, relu=False):
    """gd4d_layernorm_fwd over the last dimension of a contiguous tensor."""
    lib = _lib.load()
    c = x.shape[-1]
    out = torch.empty_like(x)
    code = lib.gd4d_layernorm_fwd(_dev(x, 'x', torch.float32), _opt(res, 'res'), _dev(gamma, 'gamma', torch.float32),
                                  _dev(beta, 'beta', torch.float32), _dev(out, 'out'), x.numel() // c, c,
                                  float(eps), int(bool(relu)), _stream())
    _lib.check(code, 'gd4d_layernorm_fwd')
    return out


def linear_ln_fwd(x, weight, bias=None, gamma=None, beta=None, eps=1e-5, x2=None, n_split=None, relu=False, r1=None,
                  r2=None, relu_after_ln=False, out=None):
    """gd4d_linear_ln_fwd: y = [ReLU] LN(act((x [+ x2]) W^T + b) + r1 + r2); gamma=None: no LayerNorm."""
    lib = _lib.load()
    k, n = x.shape[-1], weight.shape[0]
    m = x.numel() // k
    if out is None:
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    if x2 is not None and x2.shape != x.shape:
        raise ValueError('x2 must have the shape of x')
    code = lib.gd4d_linear_ln_fwd(_dev(x, 'x', torch.float32), _opt(x2, 'x2'), _dev(weight, 'weight', torch.float32),
                                  _opt(bias, 'bias'), _opt(r1, 'r1'), _opt(r2, 'r2'), _opt(gamma, 'gamma'),
                                  _opt(beta, 'beta'), _dev(out, 'out'), m, k, n, n if n_split is None else int(n_split),
                                  int(bool(relu)) | (4 if relu_after_ln else 0), float(eps), k, n, n, n, _stream())
    _lib.check(code, 'gd4d_linear_ln_fwd')
    return out


def small_linear_layernorm_fwd(x, weight, bias, gamma, beta, eps=1e-5, relu=False, inv_sigmoid_in=False):
    """gd4d_small_linear_layernorm_fwd: [ReLU] LN(f(x) W^T + b) with x (..., K <= 4) -> (..., C)."""
    lib = _lib.load()
    k, c = x.shape[-1], weight.shape[0]
    out = torch.empty(*x.shape[:-1], c, device=x.device, dtype=torch.float32)
    code = lib.gd4d_small_linear_layernorm_fwd(_dev(x, 'x', torch.float32), _dev(weight, 'weight', torch.float32),
                                               _opt(bias, 'bias'), _dev(gamma, 'gamma', torch.float32),
                                               _dev(beta, 'beta', torch.float32), _dev(out, 'out'), x.numel() // k, k,
                                               c, float(eps), int(bool(relu)) | (2 if inv_sigmoid_in else 0),
                                               _stream())
    _lib.check(code, 'gd4d_small_linear_layernorm_fwd')
    return out


def _mha_args(q, k, v, num_heads, attn_mask):
    lq, b, c = q.shape
    lk = k.shape[0]
    d = c // num_heads

    def ld(t, name):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.Gd4dError(f'{name} must be a float32 GPU tensor')
        if t.stride(2) != 1 or t.stride(0) != t.stride(1) * t.shape[1]:
            raise ValueError(f'{name} must be row-strided (L, B, C)')
        return t.stride(1)
    kind, mptr = 0, None
    if attn_mask is not None:
        if attn_mask.dim() != 2:
            raise NotImplementedError('only 2-D (Lq, Lk) attention masks are supported')
        if attn_mask.dtype in (torch.bool, torch.uint8):
            attn_mask, kind = attn_mask.to(torch.uint8), 1
        else:
            attn_mask, kind = attn_mask.float(), 2
        attn_mask = attn_mask.contiguous()
        mptr = _dev(attn_mask, 'attn_mask')
    return lq, lk, b, c, d, ld, kind, mptr, attn_mask


def _mha_seed(dropout_p, seed, device):
    if not dropout_p:
        return None
    if seed is None or seed.dtype != torch.int64 or seed.numel() < 1 or not seed.is_contiguous():
        raise ValueError('dropout_p > 0 needs seed: a contiguous int64 device tensor (mha_dropout_seed)')
    return _dev(seed, 'seed', torch.int64)


def mha_dropout_seed(device):
    """A fresh (1,) int64 device tensor holding the seed of the next dropout mask (its two 32-bit words are what
    gd4d_mha_core_fwd / _bwd read), drawn ON THE DEVICE from torch's generator for that device: torch.manual_seed makes the
    masks repeatable, and a captured launch draws a new seed on every replay (torch advances the generator's offset for
    graphs, as it does for nn.Dropout in the same step)."""
    return torch.randint(-2 ** 62, 2 ** 62, (1,), device=device, dtype=torch.int64)


def mha_dropout_keep_mask(seed, b, num_heads, lq, lk, dropout_p):
    """The (B, heads, Lq, Lk) bool keep mask gd4d_mha_core_fwd draws for this seed (csrc/gd4d_mha_dropout.h restated with
    torch integer ops): what a check against torch's softmax / bmm needs.  Not used by the product path."""
    m32 = 0xFFFFFFFF
    s = int(seed.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    lo, hi = s & m32, (s >> 32) & m32
    ids = torch.arange(b * num_heads * lq * lk, dtype=torch.int64, device=seed.device)
    x = ids ^ lo
    x = (x * 0x9E3779B1) & m32
    x = x ^ (x >> 16)
    x = (x + hi) & m32
    x = (x * 0x85EBCA6B) & m32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & m32
    x = x ^ (x >> 16)
    t = dropout_p * 4294967296.0
    thresh = 0 if t <= 0 else (m32 if t >= 4294967295.0 else int(t + 0.5))
    return (x >= thresh).view(b, num_heads, lq, lk)


def mha_core_fwd(q, k, v, num_heads, attn_mask=None, want_lse=False, dropout_p=0., seed=None):
    """gd4d_mha_core_fwd.  q (Lq, B, C), k / v (Lk, B, C): each contiguous or a last-dim slice of a packed
    (L, B, 3C) in-projection buffer.  attn_mask: None, bool/uint8 (Lq, Lk) (nonzero = masked) or float
    additive (Lq, Lk).  Returns (Lq, B, C) [, lse (Lq, B, heads) with want_lse - what mha_core_bwd needs].
    dropout_p, seed (mha_dropout_seed): dropout of the probabilities, as nn.MultiheadAttention in training."""
    lib = _lib.load()
    lq, lk, b, c, d, ld, kind, mptr, keep = _mha_args(q, k, v, num_heads, attn_mask)
    sptr = _mha_seed(dropout_p, seed, q.device)
    out = torch.empty(lq, b, c, device=q.device, dtype=torch.float32)
    lse = torch.empty(lq, b, num_heads, device=q.device, dtype=torch.float32) if want_lse else None
    code = lib.gd4d_mha_core_fwd(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()),
                                 ctypes.c_void_p(v.data_ptr()), mptr, _dev(out, 'out'), lq, lk, b,
                                 num_heads, d, ld(q, 'q'), ld(k, 'k'), ld(v, 'v'), c, kind,
                                 1.0 / (d ** 0.5), None if lse is None else _dev(lse, 'lse'), float(dropout_p), sptr,
                                 _stream())
    _lib.check(code, 'gd4d_mha_core_fwd')
    return (out, lse) if want_lse else out


class FillJob(ctypes.Structure):
    """gd4d_fill_job (include/gd4d.h)."""
    _fields_ = [('plan', ctypes.c_void_p), ('slots', ctypes.c_void_p), ('query_order', ctypes.c_void_p),
                ('id_base', ctypes.c_uint32), ('Q', ctypes.c_int32)]


def mha_core_bwd(q, k, v, out, grad_out, lse, num_heads, attn_mask=None, packed_qk=False, dropout_p=0., seed=None, fills=None):
    """gd4d_mha_core_bwd.  Returns (dq, dk, dv), each (L, B, C) contiguous; packed_qk (self-attention, q and k the two halves
    of one (L, B, 2C) projection): (dqk (L, B, 2C), dv) - the kernel writes both halves of one buffer.  dropout_p / seed:
    the forward's.  fills: what PyramidGrad.take_fills returned - one or two layers' record fills of the pyramid gradient ride
    in the dk / dv launch (gd4d_mha_core_bwd_fill)."""
    lib = _lib.load()
    lq, lk, b, c, d, ld, kind, mptr, keep = _mha_args(q, k, v, num_heads, attn_mask)
    sptr = _mha_seed(dropout_p, seed, q.device)
    f32 = torch.float32
    if packed_qk:
        if lq != lk:
            raise ValueError('packed_qk needs as many queries as keys')
        dqk = torch.empty(lq, b, 2 * c, device=q.device, dtype=f32)
        dq, dk, ldd = dqk[..., :c], dqk[..., c:], 2 * c
    else:
        dq = torch.empty(lq, b, c, device=q.device, dtype=f32)
        dk = torch.empty(lk, b, c, device=q.device, dtype=f32)
        ldd = c
    dv = torch.empty(lk, b, c, device=q.device, dtype=f32)
    dsum = torch.empty(lq, b, num_heads, device=q.device, dtype=f32)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())    # noqa: E731
    if fills is not None:
        jobs, start, records, fb, fn, fhh = fills
        arr = (FillJob * len(jobs))(*[FillJob(_dev(pl.buf, 'plan', torch.uint8).value, _dev(sl, 'slots').value,
                                               None if pl.order is None else _order_ptr(pl.order, fb * pl.q).value, int(base), int(pl.q))
                                       for pl, sl, base in jobs])
        code = lib.gd4d_mha_core_bwd_fill(vp(q), vp(k), vp(v), _dev(out, 'out', f32), _dev(grad_out, 'grad_out', f32),
                                          mptr, _dev(lse, 'lse', f32), _dev(dsum, 'dsum'), vp(dq), vp(dk),
                                          _dev(dv, 'dv'), lq, lk, b, num_heads, d, ld(q, 'q'), ld(k, 'k'), ld(v, 'v'), c, c, ldd, ldd, c,
                                          kind, 1.0 / (d ** 0.5), float(dropout_p), sptr, arr, len(jobs),
                                          _dev(start, 'start', torch.int32), _dev(records, 'records'), fb, fn, fhh, int(jobs[0][0].points), _stream())
        _lib.check(code, 'gd4d_mha_core_bwd_fill')
        return (dqk, dv) if packed_qk else (dq, dk, dv)
    code = lib.gd4d_mha_core_bwd(vp(q), vp(k), vp(v), _dev(out, 'out', f32), _dev(grad_out, 'grad_out', f32),
                                 mptr, _dev(lse, 'lse', f32), _dev(dsum, 'dsum'), vp(dq), vp(dk),
                                 _dev(dv, 'dv'), lq, lk, b, num_heads, d, ld(q, 'q'), ld(k, 'k'), ld(v, 'v'), c, c, ldd, ldd, c,
                                 kind, 1.0 / (d ** 0.5), float(dropout_p), sptr, _stream())
    _lib.check(code, 'gd4d_mha_core_bwd')
    return (dqk, dv) if packed_qk else (dq, dk, dv)


def layernorm_bwd(x, gamma, beta, grad_y, eps=1e-5, res=None, relu=False, into=None, defer=False):
    """gd4d_layernorm_bwd.  Returns (dx like x, dgamma, dbeta); into=(g_buf, b_buf): dgamma / dbeta are ADDED to these.
    defer=True: only dx and the partial column sums are computed - returns (dx, workspace, (M, C)) for
    layernorm_bwd_reduce_group."""
    lib = _lib.load()
    f32 = torch.float32
    c = x.shape[-1]
    m = x.numel() // c
    dx = torch.empty_like(x)
    dg, db = (None, None) if defer else ((torch.empty_like(gamma), torch.empty_like(gamma)) if into is None else into)
    nbytes = lib.gd4d_layernorm_bwd_workspace_bytes(m, c)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    code = lib.gd4d_layernorm_bwd(_dev(x, 'x', f32), _opt(res, 'res'), _dev(gamma, 'gamma', f32), _opt(beta, 'beta'),
                                  _dev(grad_y, 'grad_y', f32), _dev(dx, 'dx'), None if defer else _dev(dg, 'dgamma', f32),
                                  None if defer else _dev(db, 'dbeta', f32), _dev(ws, 'workspace'), ctypes.c_size_t(nbytes), m, c,
                                  float(eps), (1 if relu else 0) | (0 if into is None else 2) | (4 if defer else 0), _stream())
    _lib.check(code, 'gd4d_layernorm_bwd')
    return (dx, ws, (m, c)) if defer else (dx, dg, db)


def layernorm_bwd_reduce_group(problems, accumulate=True):
    """gd4d_layernorm_bwd_reduce_group: problems = list (<= 32) of (workspace, (M, C), dgamma, dbeta) from layernorm_bwd(defer=True)."""
    lib = _lib.load()
    n = len(problems)
    f32 = torch.float32
    arr = lambda v: (ctypes.c_void_p * n)(*v)          # noqa: E731
    dims = [int(v) for _, mc, _, _ in problems for v in mc]
    code = lib.gd4d_layernorm_bwd_reduce_group(arr([_dev(w, 'workspace').value for w, _, _, _ in problems]),
                                               arr([_dev(g, 'dgamma', f32).value for _, _, g, _ in problems]),
                                               arr([_dev(b, 'dbeta', f32).value for _, _, _, b in problems]),
                                               (ctypes.c_int32 * (2 * n))(*dims), n, 1 if accumulate else 0, _stream())
    _lib.check(code, 'gd4d_layernorm_bwd_reduce_group')


def inverse_sigmoid_fwd(x):
    """gd4d_inverse_sigmoid_fwd: the reference's inverse_sigmoid (eps = 1e-5) as one launch; no autograd."""
    lib = _lib.load()
    y = torch.empty_like(x)
    code = lib.gd4d_inverse_sigmoid_fwd(_dev(x, 'x', torch.float32), _dev(y, 'y'), x.numel(), _stream())
    _lib.check(code, 'gd4d_inverse_sigmoid_fwd')
    return y


def inverse_sigmoid_bwd(x, grad_y, add=None):
    """gd4d_inverse_sigmoid_bwd: grad_y * d inverse_sigmoid(x) / dx (+ add)."""
    lib = _lib.load()
    gx = torch.empty_like(x)
    code = lib.gd4d_inverse_sigmoid_bwd(_dev(x, 'x', torch.float32), _dev(grad_y, 'grad_y', torch.float32), _opt(add, 'add'),
                                        _dev(gx, 'grad_x'), x.numel(), _stream())
    _lib.check(code, 'gd4d_inverse_sigmoid_bwd')
    return gx


def refine_reference_fwd(tmp, ref):
    """gd4d_refine_reference_fwd: tmp (..., >=5) regression deltas, ref (..., 3) in [0,1] -> new ref."""
    lib = _lib.load()
    out = torch.empty_like(ref)
    code = lib.gd4d_refine_reference_fwd(_dev(tmp, 'tmp', torch.float32), _dev(ref, 'ref', torch.float32),
                                         _dev(out, 'out'), ref.numel() // 3, tmp.shape[-1], _stream())
    _lib.check(code, 'gd4d_refine_reference_fwd')
    return out


def frustum_pe_input_fwd(img2lidar, feat_hw, pad_hw, depth_num, depth_start, pc_range, out=None, row_start=0):
    """gd4d_frustum_pe_input_fwd.  img2lidar (R, 4, 4) fp32 -> (x, outside (R, H, W) bool).  Without `out`: x is a new
    NCHW (R, 3*D, H, W) tensor; with `out` (R, S, 3*D) the level is written channels-last at pixels
    [row_start, row_start + H*W) of every row."""
    lib = _lib.load()
    r = img2lidar.shape[0]
    h, w = feat_hw
    row_pixels = 0
    if out is None:
        out = torch.empty(r, 3 * depth_num, h, w, device=img2lidar.device, dtype=torch.float32)
    else:
        row_pixels = out.shape[1]
        if out.shape != (r, row_pixels, 3 * depth_num):
            raise ValueError('out must be (R, S, 3*D)')
    outside = torch.empty(r, h, w, device=img2lidar.device, dtype=torch.uint8)
    rng = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    code = lib.gd4d_frustum_pe_input_fwd(_dev(img2lidar, 'img2lidar', torch.float32), _dev(out, 'out', torch.float32),
                                         _dev(outside, 'outside'), r, h, w, int(depth_num), float(pad_hw[0]),
                                         float(pad_hw[1]), float(depth_start), rng, int(row_pixels), int(row_start),
                                         _stream())
    _lib.check(code, 'gd4d_frustum_pe_input_fwd')
    return out, outside.bool()


def se_fuse_chlast_fwd(feat, gate, pe, sine, row_start, out=None, out_channels_last=False):
    """gd4d_se_fuse_chlast_fwd: feat (R, C, H, W) NCHW, gate / pe (R, S, C) channels-last, sine NCHW like feat or
    channels-last like gate -> (R, C, H, W).  out_channels_last: the result's MEMORY is (R, H, W, C) - returned as its
    (R, C, H, W) view, the same values; what PyramidView.is_channels_last_level recognises and the gathers read in place."""
    lib = _lib.load()
    r, c, h, w = feat.shape
    if out is None:
        out = torch.empty((r, h, w, c) if out_channels_last else (r, c, h, w), device=feat.device, dtype=torch.float32)
    code = lib.gd4d_se_fuse_chlast_fwd(_dev(feat, 'feat', torch.float32), _dev(gate, 'gate', torch.float32),
                                       _dev(pe, 'pe', torch.float32), _dev(sine, 'sine', torch.float32), _dev(out, 'out'),
                                       r, c, h * w, gate.shape[1], int(row_start), int(sine.dim() == 3),
                                       1 if out_channels_last else 0, _stream())
    _lib.check(code, 'gd4d_se_fuse_chlast_fwd')
    return out.permute(0, 3, 1, 2) if out_channels_last else out


def sine_pe3d_fwd(n_embed, y_embed, x_embed, dim_t, out=None, row_start=0):
    """gd4d_sine_pe3d_fwd.  embeds (R, H, W) fp32, dim_t (F) -> (R, 3*F, H, W); with `out` (R, S, 3*F) the level is
    written channels-last at pixels [row_start, row_start + H*W) of every row."""
    lib = _lib.load()
    r, h, w = n_embed.shape
    f = dim_t.numel()
    row_pixels = 0
    if out is None:
        out = torch.empty(r, 3 * f, h, w, device=n_embed.device, dtype=torch.float32)
    else:
        row_pixels = out.shape[1]
    code = lib.gd4d_sine_pe3d_fwd(_dev(n_embed, 'n_embed', torch.float32), _dev(y_embed, 'y_embed', torch.float32),
                                  _dev(x_embed, 'x_embed', torch.float32), _dev(dim_t, 'dim_t', torch.float32),
                                  _dev(out, 'out', torch.float32), r, h * w, f, int(row_pixels), int(row_start), _stream())
    _lib.check(code, 'gd4d_sine_pe3d_fwd')
    return out


def se_fuse_chlast_bwd(grad_out, gate, pe, grad_sine, row_start):
    """gd4d_se_fuse_chlast_bwd for one level: grad_out (R, C, H, W); gate / pe (R, S, C) are REPLACED by their gradients on
    the level's rows, grad_sine (R, S, C) receives grad_out channels-last."""
    lib = _lib.load()
    r, c, h, w = grad_out.shape
    code = lib.gd4d_se_fuse_chlast_bwd(_dev(grad_out, 'grad_out', torch.float32), _dev(gate, 'gate', torch.float32),
                                       _dev(pe, 'pe', torch.float32), _dev(gate, 'gate'), _dev(pe, 'pe'),
                                       _dev(grad_sine, 'grad_sine', torch.float32), r, c, h * w, gate.shape[1],
                                       int(row_start), _stream())
    _lib.check(code, 'gd4d_se_fuse_chlast_bwd')


def se_fuse_fwd(feat, gate, pe, sine, out=None):
    """gd4d_se_fuse_fwd: feat + (pe * sigmoid(gate) + sine), all the same shape."""
    lib = _lib.load()
    out = torch.empty_like(feat) if out is None else out
    code = lib.gd4d_se_fuse_fwd(_dev(feat, 'feat', torch.float32), _dev(gate, 'gate', torch.float32),
                                _dev(pe, 'pe', torch.float32), _dev(sine, 'sine', torch.float32), _dev(out, 'out'),
                                ctypes.c_size_t(feat.numel()), _stream())
    _lib.check(code, 'gd4d_se_fuse_fwd')
    return out


def split_bf16_fwd(w):
    """gd4d_split_bf16_fwd: fp32 tensor -> (hi, lo) bf16 tensors of the same shape with w ~= hi + lo."""
    lib = _lib.load()
    hi = torch.empty(w.shape, device=w.device, dtype=torch.bfloat16)
    lo = torch.empty(w.shape, device=w.device, dtype=torch.bfloat16)
    code = lib.gd4d_split_bf16_fwd(_dev(w, 'w', torch.float32), _dev(hi, 'hi'), _dev(lo, 'lo'),
                                   ctypes.c_size_t(w.numel()), _stream())
    _lib.check(code, 'gd4d_split_bf16_fwd')
    return hi, lo


def gemm_bf16x3_fwd(a, w_hi, w_lo, bias=None, relu=False, out=None, relu_in=False, mask_out=False):
    """gd4d_gemm_bf16x3_fwd: a (M, K) fp32 row-major, w_hi / w_lo (N, K) bf16 -> act(a W^T + b) (M, N) fp32.
    mask_out: `out` holds a ReLU's forward output and is replaced by the result where it was > 0, by 0 elsewhere."""
    lib = _lib.load()
    m, k = a.shape
    n = w_hi.shape[0]
    if out is None:
        if mask_out:
            raise ValueError('mask_out=True needs `out` = the activations of the forward')
        out = torch.empty(m, n, device=a.device, dtype=torch.float32)
    code = lib.gd4d_gemm_bf16x3_fwd(_dev(a, 'a', torch.float32), _dev(w_hi, 'w_hi', torch.bfloat16),
                                    _dev(w_lo, 'w_lo', torch.bfloat16), _opt(bias, 'bias'), _dev(out, 'out'), m, n, k, k, n,
                                    int(bool(relu)) | (16 if relu_in else 0) | (32 if mask_out else 0), _stream())
    _lib.check(code, 'gd4d_gemm_bf16x3_fwd')
    return out


def mlp2_image(w1, b1, w2):
    """gd4d_mlp2_image: the MFMA-fragment image of a two-layer MLP's W1 (H, K1), b1 (H) or None, W2 (N2, H) for mlp2_bf16x3_fwd."""
    lib = _lib.load()
    h, k1 = w1.shape
    n2 = w2.shape[0]
    nbytes = int(lib.gd4d_mlp2_image_bytes(k1, h, n2))
    if nbytes == 0 or w2.shape[1] != h:
        raise _lib.Gd4dError(f'mlp2: K1 = {k1} (a multiple of 16, <= 256), H = {h} (a multiple of 32), N2 = {n2} (256) are the kernel\'s limits')
    img = torch.empty(nbytes, device=w1.device, dtype=torch.uint8)
    code = lib.gd4d_mlp2_image(_dev(w1.contiguous(), 'w1', torch.float32), None if b1 is None else _dev(b1.contiguous(), 'b1', torch.float32),
                               _dev(w2.contiguous(), 'w2', torch.float32), k1, h, n2, _dev(img, 'image', torch.uint8), _stream())
    _lib.check(code, 'gd4d_mlp2_image')
    img.shape_khn = (k1, h, n2)
    return img


def mlp2_supported(k1, h, n2):
    return int(_lib.load().gd4d_mlp2_image_bytes(int(k1), int(h), int(n2))) > 0


def mlp2_bf16x3_fwd(x, image, b2=None, out=None):
    """gd4d_mlp2_bf16x3_fwd: x (M, K1) fp32 -> relu(x W1^T + b1) W2^T + b2 (M, N2), the hidden activation never stored."""
    lib = _lib.load()
    k1, h, n2 = image.shape_khn
    m = x.shape[0]
    if x.shape[1] != k1:
        raise ValueError(f'mlp2_bf16x3_fwd: x has {x.shape[1]} columns, the image was made for {k1}')
    if out is None:
        out = torch.empty(m, n2, device=x.device, dtype=torch.float32)
    code = lib.gd4d_mlp2_bf16x3_fwd(_dev(x, 'x', torch.float32), _dev(image, 'image', torch.uint8), _opt(b2, 'b2'),
                                    _dev(out, 'out', torch.float32), m, k1, h, n2, k1, n2, _stream())
    _lib.check(code, 'gd4d_mlp2_bf16x3_fwd')
    return out


def mlp2_pe_se_fwd(img2lidar, feats, pad_hw, depth_num, depth_start, pc_range, pe_image, pe_b2, se_image, se_b2, sine, outs=None,
                   pe_out=None):
    """gd4d_mlp2_pe_se_fwd: position_encoder(frustum) and the SE gate + fuse in one kernel - mlp2_frustum_fwd + mlp2_se_fuse_fwd without
    the (R, S, 256) embedding between them.  feats: L levels (R, 256, H_l, W_l) NCHW of the cameras of img2lidar (R, 4, 4); sine
    (R, S, 256); outs: L (R, H_l, W_l, 256) tensors to write (made when None); pe_out (R, S, 256): store the embedding as well.
    Returns the levels as (R, 256, H_l, W_l) views of channels-last memory."""
    lib = _lib.load()
    f32 = torch.float32
    k1, pe_h, n2 = pe_image.shape_khn
    sk, se_h, sn = se_image.shape_khn
    nl = len(feats)
    r = img2lidar.shape[0]
    if k1 != 3 * depth_num or n2 != 256 or sk != 256 or sn != 256 or any(f.shape[0] != r or f.shape[1] != 256 or f.dim() != 4 for f in feats):
        raise ValueError('mlp2_pe_se_fwd: a 3 D -> H -> 256 image, a 256 -> H -> 256 image and (R, 256, H, W) levels expected')
    s_tot = sum(f.shape[2] * f.shape[3] for f in feats)
    if sine.numel() != r * s_tot * 256 or (pe_out is not None and pe_out.numel() != r * s_tot * 256):
        raise ValueError(f'mlp2_pe_se_fwd: sine / pe_out must hold ({r}, {s_tot}, 256)')
    if outs is None:
        outs = [torch.empty(r, f.shape[2], f.shape[3], 256, device=f.device, dtype=f32) for f in feats]
    elif any(tuple(o.shape) != (r, f.shape[2], f.shape[3], 256) for o, f in zip(outs, feats)):
        raise ValueError('mlp2_pe_se_fwd: outs must be (R, H_l, W_l, 256) per level')
    fp = (ctypes.c_void_p * nl)(*[_dev(f, 'feats', f32).value for f in feats])
    op = (ctypes.c_void_p * nl)(*[_dev(o, 'outs', f32).value for o in outs])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[2:]])
    rng = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    code = lib.gd4d_mlp2_pe_se_fwd(_dev(img2lidar, 'img2lidar', f32), fp, lv, nl, r, float(pad_hw[0]), float(pad_hw[1]), int(depth_num),
                                   float(depth_start), rng, _dev(pe_image, 'pe_image', torch.uint8), _opt(pe_b2, 'pe_b2'), pe_h,
                                   _dev(se_image, 'se_image', torch.uint8), _opt(se_b2, 'se_b2'), se_h, _dev(sine, 'sine', f32), op,
                                   None if pe_out is None else _dev(pe_out, 'pe_out', f32), _stream())
    _lib.check(code, 'gd4d_mlp2_pe_se_fwd')
    return [o.permute(0, 3, 1, 2) for o in outs]


def mlp2_se_fuse_fwd(feats, image, b2, pe, sine, outs=None):
    """gd4d_mlp2_se_fuse_fwd: feats = L levels (R, 256, H_l, W_l) NCHW, image = mlp2_image(conv_reduce.weight, conv_reduce.bias,
    conv_expand.weight), b2 = conv_expand.bias, pe / sine (R, S, 256) channels-last rows of all levels side by side ->
    L tensors feat + (pe * sigmoid(gate) + sine) as (R, 256, H_l, W_l) VIEWS of (R, H_l, W_l, 256) memory (channels-last levels;
    outs: the L (R, H_l, W_l, 256) tensors to write)."""
    lib = _lib.load()
    k1, h, n2 = image.shape_khn
    nl = len(feats)
    r = feats[0].shape[0]
    f32 = torch.float32
    if k1 != 256 or n2 != 256 or any(f.shape[0] != r or f.shape[1] != 256 or f.dim() != 4 for f in feats):
        raise ValueError('mlp2_se_fuse_fwd: (R, 256, H, W) levels and a 256 -> H -> 256 image expected')
    s_tot = sum(f.shape[2] * f.shape[3] for f in feats)
    if tuple(pe.shape) != (r, s_tot, 256) or tuple(sine.shape) != (r, s_tot, 256):
        raise ValueError(f'mlp2_se_fuse_fwd: pe / sine must be ({r}, {s_tot}, 256)')
    if outs is None:
        outs = [torch.empty(r, f.shape[2], f.shape[3], 256, device=f.device, dtype=f32) for f in feats]
    elif any(tuple(o.shape) != (r, f.shape[2], f.shape[3], 256) for o, f in zip(outs, feats)):
        raise ValueError('mlp2_se_fuse_fwd: outs must be (R, H_l, W_l, 256) per level')
    fp = (ctypes.c_void_p * nl)(*[_dev(f, 'feats', f32).value for f in feats])
    op = (ctypes.c_void_p * nl)(*[_dev(o, 'outs', f32).value for o in outs])
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for f in feats for x in f.shape[2:]])
    code = lib.gd4d_mlp2_se_fuse_fwd(fp, lv, nl, r, _dev(image, 'image', torch.uint8), _opt(b2, 'b2'), _dev(pe, 'pe', f32),
                                     _dev(sine, 'sine', f32), op, 256, h, _stream())
    _lib.check(code, 'gd4d_mlp2_se_fuse_fwd')
    return [o.permute(0, 3, 1, 2) for o in outs]


def mlp2_frustum_image(w1, b1, w2):
    """The image gd4d_mlp2_frustum_fwd takes: mlp2_image of W1 (H, 192) with its columns in the order the kernel's lanes generate the
    frustum inputs in - column 16 st + 8 kg + e of the image's W1 = column 96 kg + 8 st + e of the module's."""
    if w1.shape[1] != 192:
        raise _lib.Gd4dError('mlp2_frustum_image: position_encoder[0] must take 3 x 64 depth bins')
    col = torch.arange(192, device=w1.device)
    st, kg, e = col // 16, (col % 16) // 8, col % 8
    return mlp2_image(w1[:, 96 * kg + 8 * st + e].contiguous(), b1, w2)


def mlp2_frustum_fwd(img2lidar, level_hw, pad_hw, depth_num, depth_start, pc_range, image, b2=None, out=None):
    """gd4d_mlp2_frustum_fwd: img2lidar (R, 4, 4) -> position_encoder(frustum coordinates) (R, S, 256), S = the pixels of `level_hw`'s
    levels side by side; image: mlp2_frustum_image.  No (R, S, 192) frustum tensor is written or read."""
    lib = _lib.load()
    k1, h, n2 = image.shape_khn
    r = img2lidar.shape[0]
    nl = len(level_hw)
    s_tot = sum(int(a) * int(b) for a, b in level_hw)
    if k1 != 3 * depth_num or n2 != 256:
        raise ValueError(f'mlp2_frustum_fwd: the image is {k1} -> {h} -> {n2}, the frustum has {3 * depth_num} channels')
    if out is None:
        out = torch.empty(r, s_tot, n2, device=img2lidar.device, dtype=torch.float32)
    elif out.numel() != r * s_tot * n2:
        raise ValueError('mlp2_frustum_fwd: out must hold (R, S, 256)')
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    rng = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    code = lib.gd4d_mlp2_frustum_fwd(_dev(img2lidar, 'img2lidar', torch.float32), lv, nl, r, float(pad_hw[0]), float(pad_hw[1]),
                                     int(depth_num), float(depth_start), rng, _dev(image, 'image', torch.uint8), _opt(b2, 'b2'),
                                     _dev(out, 'out', torch.float32), h, n2, _stream())
    _lib.check(code, 'gd4d_mlp2_frustum_fwd')
    return out.view(r, s_tot, n2)


_TN_WS = {}


def gemm_tn_bf16x3(a, b, relu_b=False, want_colsum=True):
    """gd4d_gemm_tn_bf16x3: a (R, M), b (R, N) fp32 row-major -> (a^T b (M, N), column sums of a (M) or None): the weight /
    bias gradients of a Linear over R rows (a = output gradient, b = input; relu_b: ReLU on b as it is read)."""
    lib = _lib.load()
    r, m = a.shape
    n = b.shape[1]
    if b.shape[0] != r:
        raise ValueError(f'gemm_tn_bf16x3: {tuple(a.shape)} against {tuple(b.shape)}')
    dev = a.device
    nbytes = lib.gd4d_gemm_tn_bf16x3_workspace_bytes(r, m, n)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _TN_WS[key] = ws
    c = torch.empty(m, n, device=dev, dtype=torch.float32)
    col = torch.empty(m, device=dev, dtype=torch.float32) if want_colsum else None
    code = lib.gd4d_gemm_tn_bf16x3(_dev(a, 'a', torch.float32), _dev(b, 'b', torch.float32), _dev(c, 'c'), _opt(col, 'colsum'),
                                   _dev(ws, 'workspace'), r, m, n, m, n, 16 if relu_b else 0, _stream())
    _lib.check(code, 'gd4d_gemm_tn_bf16x3')
    return c, col


def knn_farthest_fwd(x, k):
    """gd4d_knn_farthest_fwd: x (B, N, C) fp32 -> (B, N, K) int32 indices of the K farthest rows of the same sample."""
    lib = _lib.load()
    b, n, c = x.shape
    idx = torch.empty(b, n, k, device=x.device, dtype=torch.int32)
    code = lib.gd4d_knn_farthest_fwd(_dev(x, 'x', torch.float32), _dev(idx, 'idx'), b, n, c, int(k), _stream())
    _lib.check(code, 'gd4d_knn_farthest_fwd')
    return idx


def edge_conv_max_fwd(ab, idx, scale, shift):
    """gd4d_edge_conv_max_fwd: ab (B, N, 2C) = [W_a x | W_b x], idx (B, N, K) int32, scale / shift (C) -> (B, N, C)."""
    lib = _lib.load()
    b, n, c2 = ab.shape
    c = c2 // 2
    out = torch.empty(b, n, c, device=ab.device, dtype=torch.float32)
    base = _dev(ab, 'ab', torch.float32)
    code = lib.gd4d_edge_conv_max_fwd(base, ctypes.c_void_p(ab.data_ptr() + 4 * c), _dev(idx, 'idx', torch.int32),
                                      _dev(scale, 'scale', torch.float32), _dev(shift, 'shift', torch.float32),
                                      _dev(out, 'out'), b, n, c, idx.shape[-1], c2, _stream())
    _lib.check(code, 'gd4d_edge_conv_max_fwd')
    return out


def box_head_fwd(tmp, ref, pc_range, scale=1.0, out=None):
    """gd4d_box_head_fwd: tmp (..., code) raw regression output, ref (..., 3) in [0,1] -> bbox_preds."""
    lib = _lib.load()
    out = torch.empty_like(tmp) if out is None else out
    rng = (ctypes.c_double * 6)(*[float(v) for v in pc_range])
    code = lib.gd4d_box_head_fwd(_dev(tmp, 'tmp', torch.float32), _dev(ref, 'ref', torch.float32), rng,
                                 float(scale), _dev(out, 'out', torch.float32), ref.numel() // 3, tmp.shape[-1],
                                 _stream())
    _lib.check(code, 'gd4d_box_head_fwd')
    return out


def nms_free_decode_fwd(cls_scores, bbox_preds, post_center_range, max_num, score_threshold=None):
    """gd4d_nms_free_decode_fwd.  cls_scores (B, Q, C) logits, bbox_preds (B, Q, code).
    Returns boxes (B, K, 9|7), scores (B, K), labels (B, K) int32, keep (B, K) bool, all sorted by score."""
    lib = _lib.load()
    b, q, c = cls_scores.shape
    code_size = bbox_preds.shape[-1]
    k = int(max_num)
    if k > q * c:
        raise RuntimeError('selected index k out of range')        # what torch.topk raises in the reference
    dev = cls_scores.device
    boxes = torch.empty(b, k, 9 if code_size > 8 else 7, device=dev, dtype=torch.float32)
    scores = torch.empty(b, k, device=dev, dtype=torch.float32)
    labels = torch.empty(b, k, device=dev, dtype=torch.int32)
    keep = torch.empty(b, k, device=dev, dtype=torch.uint8)
    rng = (ctypes.c_float * 6)(*[float(v) for v in post_center_range])
    thr = -1.0 if score_threshold is None else float(score_threshold)
    code = lib.gd4d_nms_free_decode_fwd(_dev(cls_scores, 'cls_scores', torch.float32),
                                        _dev(bbox_preds, 'bbox_preds', torch.float32), rng, thr, _dev(boxes, 'boxes'),
                                        _dev(scores, 'scores'), _dev(labels, 'labels'), _dev(keep, 'keep'),
                                        b, q, c, code_size, k, _stream())
    _lib.check(code, 'gd4d_nms_free_decode_fwd')
    return boxes, scores, labels, keep.bool()


def refine_reference_order_fwd(tmp, ref, pc_range):
    """gd4d_refine_reference_order_fwd: tmp (B, Q, >=5), ref (B, Q, 3) -> (new ref, int32 locality order of it)."""
    lib = _lib.load()
    b, q = ref.shape[0], ref.shape[1]
    out = torch.empty_like(ref)
    order = torch.empty(b * q, device=ref.device, dtype=torch.int32)
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    code = lib.gd4d_refine_reference_order_fwd(_dev(tmp, 'tmp', torch.float32), _dev(ref, 'ref', torch.float32),
                                               _dev(out, 'out'), rng, _dev(order, 'order'), b, q, tmp.shape[-1],
                                               _stream())
    _lib.check(code, 'gd4d_refine_reference_order_fwd')
    return out, order


def cross_attn_bwd(value, level_hw, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w,
                   grad_out, query_order=None, raw_cam_weights=False):
    """gd4d_cross_attn_bwd.  Returns (grad_value, grad_ref, grad_offsets, grad_attn_logits, grad_cam_logits)."""
    lib = _lib.load()
    f32 = torch.float32
    b, q = ref.shape[0], ref.shape[1]
    n = lidar2img.shape[1]
    hh, dh = value.shape[2], value.shape[3]
    p = offsets.shape[3]
    nl = len(level_hw)
    gv = torch.zeros_like(value, dtype=f32)
    gr = torch.empty_like(ref)
    go = torch.empty(b, q, hh, p, 3, device=ref.device, dtype=f32)
    ga = torch.empty(b, q, hh, nl, p, device=ref.device, dtype=f32)
    gc = torch.empty(b, q, n, device=ref.device, dtype=f32)
    lv = (ctypes.c_int32 * (2 * nl))(*[int(x) for hw in level_hw for x in hw])
    rng = (ctypes.c_double * 6)(*[float(x) for x in pc_range])
    nbytes = lib.gd4d_cross_attn_bwd_workspace_bytes(b, q, hh, nl, p)          # B > 1: partial logit gradients per sample
    ws = torch.empty(nbytes, device=ref.device, dtype=torch.uint8) if nbytes else None
    code = lib.gd4d_cross_attn_bwd(
        _dev(value, 'value', f32), lv, _dev(ref, 'ref', f32), _dev(offsets, 'offsets', f32),
        _dev(attn_logits, 'attn_logits', f32), _dev(cam_logits, 'cam_logits', f32),
        _dev(lidar2img, 'lidar2img', f32), rng, float(img_h), float(img_w), _dev(grad_out, 'grad_out', f32),
        _dev(gv, 'grad_value'), _dev(gr, 'grad_ref'), _dev(go, 'grad_offsets'), _dev(ga, 'grad_attn_logits'),
        _dev(gc, 'grad_cam_logits'), b, n, q, hh, dh, nl, p, _lib.F32, _lib.PIXEL_MAJOR, 1 if raw_cam_weights else 0,
        None if query_order is None else _order_ptr(query_order, b * q),
        None if ws is None else _dev(ws, 'workspace'), ctypes.c_size_t(nbytes), _stream())
    _lib.check(code, 'gd4d_cross_attn_bwd')
    return gv, gr, go, ga, gc


def match_cost_fwd(cls, box, gt_boxes, gt_labels, gt_start, max_gt, cls_weight=2.0, reg_weight=0.25, alpha=0.25):
    """gd4d_match_cost_fwd.  cls (NL, B, Q, C), box (NL, B, Q, code) fp32; gt_boxes (sumG, 7..9) fp32, gt_labels (sumG)
    int32, gt_start (B + 1) int32 - all on the GPU; max_gt = largest per-sample count.  Returns the flat cost buffer
    (NL * Q * sumG): block (l, b) at Q * (l * sumG + gt_start[b]), shape (Q, G_b)."""
    lib = _lib.load()
    f32, i32 = torch.float32, torch.int32
    nl, b, q, c = cls.shape
    sum_gt = gt_boxes.shape[0]
    cost = torch.empty(nl * q * sum_gt, device=cls.device, dtype=f32)
    code = lib.gd4d_match_cost_fwd(_dev(cls, 'cls', f32), _dev(box, 'box', f32), _dev(gt_boxes, 'gt_boxes', f32),
                                   _dev(gt_labels, 'gt_labels', i32), _dev(gt_start, 'gt_start', i32),
                                   _dev(cost, 'cost'), nl, b, q, c, box.shape[-1], gt_boxes.shape[-1], sum_gt,
                                   int(max_gt), float(cls_weight), float(reg_weight), float(alpha), _stream())
    _lib.check(code, 'gd4d_match_cost_fwd')
    return cost


def head_loss_fwd_bwd(cls, box, assigned, gt_boxes, gt_labels, code_weights, avg_factors, alpha=0.25,
                      loss_cls_weight=2.0, loss_bbox_weight=0.25):
    """gd4d_head_loss_fwd_bwd.  Returns (loss (NL, 2), grad_cls like cls, grad_box like box)."""
    lib = _lib.load()
    f32, i32 = torch.float32, torch.int32
    nl, b, q, c = cls.shape
    loss = torch.empty(nl, 2, device=cls.device, dtype=f32)
    gcls, gbox = torch.empty_like(cls), torch.empty_like(box)
    code = lib.gd4d_head_loss_fwd_bwd(_dev(cls, 'cls', f32), _dev(box, 'box', f32), _dev(assigned, 'assigned', i32),
                                      _dev(gt_boxes, 'gt_boxes', f32), _dev(gt_labels, 'gt_labels', i32),
                                      _dev(code_weights, 'code_weights', f32), _dev(avg_factors, 'avg_factors', f32),
                                      _dev(loss, 'loss'), _dev(gcls, 'grad_cls'), _dev(gbox, 'grad_box'), nl, b, q, c,
                                      box.shape[-1], gt_boxes.shape[-1], gt_boxes.shape[0], float(alpha),
                                      float(loss_cls_weight), float(loss_bbox_weight), _stream())
    _lib.check(code, 'gd4d_head_loss_fwd_bwd')
    return loss, gcls, gbox


def linear_sum_assignment_batch(cost, problems, num_threads=8):
    """gd4d_linear_sum_assignment_batch (host).  cost: contiguous float32 numpy array; problems: list of
    (offset, rows, cols) into it.  Returns a list of int32 arrays (rows,): assigned column per row or -1."""
    import numpy as np
    lib = _lib.load()
    cost = np.ascontiguousarray(cost, dtype=np.float32)
    n = len(problems)
    offs = np.asarray([p[0] for p in problems], dtype=np.int64)
    rows = np.asarray([p[1] for p in problems], dtype=np.int32)
    cols = np.asarray([p[2] for p in problems], dtype=np.int32)
    if n and int((offs + rows.astype(np.int64) * cols).max()) > cost.size:
        raise ValueError('a problem reaches past the end of the cost buffer')
    out_off = np.concatenate([[0], np.cumsum(rows, dtype=np.int64)]).astype(np.int64)
    out = np.empty(int(out_off[-1]), dtype=np.int32)
    as_p = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    code = lib.gd4d_linear_sum_assignment_batch(as_p(cost), as_p(offs), as_p(rows), as_p(cols), n, as_p(out),
                                                as_p(out_off), int(num_threads))
    _lib.check(code, 'gd4d_linear_sum_assignment_batch')
    return [out[out_off[i]:out_off[i + 1]] for i in range(n)]


def hungarian_assign_fwd(cost, gt_start, nl, b, q, sum_gt, max_gt, assigned=None, status=None, workspace=None):
    """gd4d_hungarian_assign_fwd: the assignment of every (layer, sample) block of match_cost_fwd's buffer on the device.  Returns
    (assigned (NL, B, Q) int32: index into the packed ground truth or -1, status (NL * B) int32: 0 solved / 1 NaN cost (bad label) /
    2 infeasible).  No host synchronisation."""
    lib = _lib.load()
    dev = cost.device
    i32 = torch.int32
    if assigned is None:
        assigned = torch.empty(nl, b, q, device=dev, dtype=i32)
    if status is None:
        status = torch.empty(nl * b, device=dev, dtype=i32)
    nbytes = int(lib.gd4d_hungarian_assign_workspace_bytes(nl, b, q, max(int(max_gt), 1)))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    code = lib.gd4d_hungarian_assign_fwd(_dev(cost, 'cost', torch.float32), _dev(gt_start, 'gt_start', i32), _dev(assigned, 'assigned', i32),
                                         _dev(status, 'status', i32), _dev(workspace, 'workspace', torch.uint8), workspace.numel(),
                                         int(nl), int(b), int(q), int(sum_gt), int(max_gt), _stream())
    _lib.check(code, 'gd4d_hungarian_assign_fwd')
    return assigned, status


class ChainOp(ctypes.Structure):
    """gd4d_chain_op (include/gd4d.h)."""
    _fields_ = [('kind', ctypes.c_int32), ('src', ctypes.c_int32), ('dst', ctypes.c_int32), ('res', ctypes.c_int32),
                ('K', ctypes.c_int32), ('N', ctypes.c_int32), ('flags', ctypes.c_int32), ('dst_col', ctypes.c_int32),
                ('ld0', ctypes.c_int32), ('ld1', ctypes.c_int32), ('ld2', ctypes.c_int32), ('ldg', ctypes.c_int32),
                ('eps', ctypes.c_float), ('reserved', ctypes.c_int32),
                ('p0', ctypes.c_void_p), ('p1', ctypes.c_void_p), ('p2', ctypes.c_void_p), ('gout', ctypes.c_void_p),
                ('p3', ctypes.c_void_p)]


CHAIN_LOAD, CHAIN_GEMM, CHAIN_LAYERNORM, CHAIN_ADD, CHAIN_REFINE, CHAIN_SMALL_LINEAR, CHAIN_HEADGEMM, CHAIN_SIGNAL, CHAIN_WAIT = 1, 2, 3, 4, 5, 6, 7, 8, 9
CHAIN_LN_BWD, CHAIN_DROPMASK = 10, 11
CHAIN_RELU, CHAIN_INV_SIGMOID, CHAIN_SIGMOID, CHAIN_EXACT, CHAIN_SRC2, CHAIN_SPLIT_OUT, CHAIN_MASK_P2, CHAIN_DROPOUT = 1, 2, 4, 8, 16, 32, 64, 128
CHAIN_SPLIT_KV, CHAIN_SPLIT_KV_KEEP, CHAIN_ADD_GOUT = 256, 512, 1024


def _rows(t, name):
    """(data pointer, row stride in elements) of a fp32 GPU tensor whose last dimension is dense and whose leading
    dimensions collapse to rows of one stride (a contiguous tensor or a last-dim slice of one)."""
    if t is None:
        return None, 0
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.Gd4dError(f'{name} must be a float32 GPU tensor')
    if t.dim() == 1:
        return t.data_ptr(), 0
    if t.stride(-1) != 1:
        raise ValueError(f'{name}: the last dimension must be dense')
    ld = t.stride(-2)
    for d in range(t.dim() - 2):
        if t.shape[d] != 1 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
            raise ValueError(f'{name}: rows must have one stride')
    return t.data_ptr(), ld


def chain_load(dst, x, x2=None, dst_col=0, inv_sigmoid=False, out=None):
    """buf[dst][:, dst_col ..] = f(x) (+ x2); out: the rows are also stored there."""
    p0, ld0 = _rows(x, 'x')
    p1, ld1 = _rows(x2, 'x2')
    g, ldg = _rows(out, 'out')
    return ChainOp(kind=CHAIN_LOAD, src=-1, dst=dst, res=-1, N=x.shape[-1], dst_col=dst_col, ld0=ld0, ld1=ld1, ldg=ldg,
                   flags=CHAIN_INV_SIGMOID if inv_sigmoid else 0, p0=p0, p1=p1, gout=g)


_CHAIN_IMAGES = {}
_CHAIN_EPOCH = [0]


def invalidate_chain_images():
    """Forget every cached weight image.  The cache notices re-assignment and in-place autograd-visible writes (a tensor's
    version counter), but NOT writes through `.data` (p.data.copy_(), mmcv's EMA swap): call this after such an update -
    the package's modules do it from train() / eval() and after load_state_dict."""
    _CHAIN_IMAGES.clear()
    _VP_IMAGES.clear()
    _CHAIN_EPOCH[0] += 1


def chain_weight_image(weight, exact=False):
    """The bf16 hi / lo (exact: hi / mid / lo) MFMA-fragment image of a (N, K) fp32 weight (gd4d_chain_weight_image[_exact]),
    cached while the weight tensor (address, shape, version counter) does not change - one small launch after a
    load_state_dict or an optimizer step, none in steady-state inference.  Entries die with their tensor."""
    import weakref
    if not weight.is_cuda or weight.dtype != torch.float32 or weight.dim() != 2 or weight.stride(1) != 1 \
            or weight.stride(0) != weight.shape[1]:
        raise ValueError('chain weights must be dense (N, K) float32 GPU tensors')
    base = weight._base if weight._base is not None else weight
    key = (weight.data_ptr(), tuple(weight.shape), bool(exact))
    hit = _CHAIN_IMAGES.get(key)
    if hit is not None and hit[0]() is base and hit[1] == base._version:
        return hit[2]
    lib = _lib.load()
    n, k = weight.shape
    nbytes = (lib.gd4d_chain_weight_image_exact_bytes if exact else lib.gd4d_chain_weight_image_bytes)(n, k)
    if nbytes == 0:
        raise _lib.Gd4dError(f'chain GEMM: K = {k} must be a multiple of 64')
    img = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    with torch.cuda.device(weight.device):
        fn = lib.gd4d_chain_weight_image_exact if exact else lib.gd4d_chain_weight_image
        code = fn(ctypes.c_void_p(weight.data_ptr()), n, k, ctypes.c_void_p(img.data_ptr()), _stream())
    _lib.check(code, 'gd4d_chain_weight_image')
    _CHAIN_IMAGES[key] = (weakref.ref(base, lambda _r, key=key: _CHAIN_IMAGES.pop(key, None)), base._version, img)
    return img


class ImageJob(ctypes.Structure):
    """gd4d_image_job (include/gd4d.h)."""
    _fields_ = [('seg', ctypes.c_void_p * 3), ('rows', ctypes.c_int32 * 3), ('cols', ctypes.c_int32), ('transposed', ctypes.c_int32),
                ('planes', ctypes.c_int32), ('frag0', ctypes.c_int32), ('reserved', ctypes.c_int32), ('image', ctypes.c_void_p)]


class WeightImage:
    """A chain GEMM operand made by an ImageSet: the image tensor and the (N, K) of the GEMM it serves (K already padded)."""

    def __init__(self, img, n, k, exact):
        self.img, self.n, self.k, self.exact = img, n, k, exact


class ImageSet:
    """The weight images of a TRAINING step's chains (gd4d_chain_weight_image_group): declared once (add / add_concat; the
    parameters' storage must stay where it is), rebuilt by ONE launch per step (refresh) into buffers whose addresses never
    change - a captured hipGraph replays the rebuild with the step, after every optimizer update."""

    def __init__(self, device):
        self.device = device
        self._jobs, self._keep, self._frags = [], [], 0
        self._table = None
        self.sources = []

    def _segments(self, tensors, cols):
        if not 1 <= len(tensors) <= 3:
            raise ValueError('an image stacks one to three row blocks')
        seg, rows = (ctypes.c_void_p * 3)(), (ctypes.c_int32 * 3)()
        for i, t in enumerate(tensors):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() % cols:
                raise ValueError('image sources must be dense float32 GPU tensors of `cols` columns')
            seg[i], rows[i] = t.data_ptr(), t.numel() // cols
            self.sources.append(t)
        return seg, rows, int(sum(rows))

    def add(self, tensors, transposed=False, exact=False):
        """Image of the row blocks `tensors` (each (r_i, cols)) stacked - or of the stack's transpose."""
        cols = tensors[0].shape[-1]
        seg, rows, r = self._segments(tensors, cols)
        n, k = (cols, r) if transposed else (r, cols)
        kp = (k + 63) // 64 * 64
        planes = 3 if exact else 2
        frags = (n + 15) // 16 * (kp // 32)
        img = torch.empty(frags * planes * 1024, device=self.device, dtype=torch.uint8)
        self._jobs.append(ImageJob(seg=seg, rows=rows, cols=cols, transposed=int(transposed), planes=planes, frag0=self._frags,
                                   image=img.data_ptr()))
        self._frags += frags
        self._table = None
        return WeightImage(img, n, kp, exact)

    def add_concat(self, vectors):
        """fp32 concatenation of 1-D tensors (a stacked bias), refreshed with the images."""
        seg, rows, r = self._segments(vectors, 1)
        out = torch.empty(r, device=self.device, dtype=torch.float32)
        self._jobs.append(ImageJob(seg=seg, rows=rows, cols=1, transposed=0, planes=0, frag0=self._frags, image=out.data_ptr()))
        self._frags += (r + 63) // 64
        self._table = None
        return out

    def signature(self):
        return tuple(t.data_ptr() for t in self.sources)

    def finalize(self):
        """Upload the job table (a host-to-device copy: call it outside a graph capture; refresh() does it on first use)."""
        if self._table is None:
            arr = (ImageJob * len(self._jobs))(*self._jobs)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._table = raw.to(self.device)

    def refresh(self):
        lib = _lib.load()
        self.finalize()
        if len(self._jobs) > 1024:
            raise _lib.Gd4dError('ImageSet: more than 1024 jobs (GD4D_IMAGE_JOBS_MAX) - split the set')
        with torch.cuda.device(self.device):
            code = lib.gd4d_chain_weight_image_group(ctypes.c_void_p(self._table.data_ptr()), len(self._jobs), self._frags, _stream())
        _lib.check(code, 'gd4d_chain_weight_image_group')


def _image_of(weight, exact=False):
    """(image pointer, N, K) of a chain GEMM operand: a WeightImage of an ImageSet, or a weight tensor (cached image)."""
    if isinstance(weight, WeightImage):
        if weight.exact != bool(exact):
            raise ValueError('chain GEMM: the image was made for the other arithmetic (exact)')
        return weight.img.data_ptr(), weight.n, weight.k
    return chain_weight_image(weight, exact).data_ptr(), weight.shape[0], weight.shape[1]


def chain_dropout_args(p):
    """(threshold, scale) of a dropout with probability p as the chain operations take them (gd4d_mha_dropout.h)."""
    t = float(p) * 4294967296.0
    thresh = 0 if t <= 0 else (0xFFFFFFFF if t >= 4294967295.0 else int(t + 0.5))
    return thresh, 1.0 / (1.0 - float(p))


def chain_dropout_keep_mask(seed, m, n, p):
    """The (m, n) bool keep mask a chain GEMM with dropout=(seed, p) and N = n draws (restated with torch integer ops, as
    mha_dropout_keep_mask): for tests against torch with the same mask."""
    return mha_dropout_keep_mask(seed, 1, 1, m, n, p).view(m, n)


def chain_dropmask(src, dst, n, seed, p, out=None):
    """buf[dst] = (the keep mask of a forward GEMM with dropout=(seed, p), N = n) * buf[src] / (1 - p); out: also stored."""
    thresh, scale = chain_dropout_args(p)
    g, ldg = _rows(out, 'out')
    c_thresh = thresh - (1 << 32) if thresh >= (1 << 31) else thresh
    return ChainOp(kind=CHAIN_DROPMASK, src=src, dst=dst, res=-1, N=n, eps=scale, reserved=c_thresh, ldg=ldg, gout=g,
                   p0=_dev(seed, 'seed', torch.int64).value)


# Measurement switch (bench.py's `exact_gemms` figure; `with ops.all_exact():`): EVERY GEMM operation of the chains built while it is
# on uses six bf16 products (GD4D_CHAIN_EXACT, ~2^-24) - what fp32-class arithmetic on the query side costs.  HEADGEMM (value_proj
# of the aggregates) and the attention core's two products stay on three.
ALL_EXACT = [os.environ.get('GD4D_CHAIN_ALL_EXACT') == '1']


class all_exact:
    def __enter__(self):
        self.prev, ALL_EXACT[0] = ALL_EXACT[0], True

    def __exit__(self, *exc):
        ALL_EXACT[0] = self.prev


def chain_gemm(src, weight, bias=None, dst=-1, dst_col=0, relu=False, res=-1, out=None, sigmoid=False, exact=False, add=None,
               add2=None, mask=None, mask_scale=0., dropout=None):
    """act(buf[src] W^T + b) (+ buf[res]) (+ (add + add2)[m, :]) -> buf[dst] and / or out.  weight (N, K) contiguous rows.
    add / add2: global (M, N) tensors added in the epilogue (their sum first, then onto the result - what a LOAD of
    add + add2 into buf[res] would give, without the operation).  exact: fp32-class products (GD4D_CHAIN_EXACT) instead of
    split-bf16 x3 - for outputs that become reference points."""
    exact = exact or ALL_EXACT[0]
    img, n, k = _image_of(weight, exact)
    g, ldg = _rows(out, 'out')
    if mask is not None and (add is not None or add2 is not None):
        raise ValueError('chain_gemm: mask (a backward chain\'s ReLU) and global addends exclude each other')
    p2, ld2 = _rows(add if mask is None else mask, 'add')
    p3, ld3 = _rows(add2, 'add2')
    if p3 is not None and p2 is None:
        raise ValueError('chain_gemm: add2 without add')
    eps, reserved, flags = float(mask_scale) if mask is not None else 0., 0, 0
    if dropout is not None and float(dropout[1]) > 0.:      # (seed: (1,) int64 device tensor, p): nn.Dropout on the output
        if mask is not None or p3 is not None:
            raise ValueError('chain_gemm: dropout excludes mask and add2')
        thresh, eps = chain_dropout_args(dropout[1])
        reserved = thresh - (1 << 32) if thresh >= (1 << 31) else thresh
        p3, flags = _dev(dropout[0], 'seed', torch.int64).value, CHAIN_DROPOUT
    return ChainOp(kind=CHAIN_GEMM, src=src, dst=dst, res=res, K=k, N=n, dst_col=dst_col,
                   flags=(CHAIN_RELU if relu else 0) | (CHAIN_SIGMOID if sigmoid else 0) | (CHAIN_EXACT if exact else 0) |
                   (CHAIN_MASK_P2 if mask is not None else 0) | flags, ldg=ldg, eps=eps, reserved=reserved,
                   ld2=ld2, ld1=ld3, p0=img, p1=None if bias is None else bias.data_ptr(), p2=p2, p3=p3, gout=g)


class KVPlanes:
    """The attention core's K / V operands as split-bf16 planes in its MFMA fragment layout (GD4D_CHAIN_SPLIT_KV ->
    gd4d_mha_core_presplit_fwd; include/gd4d.h has the layouts): k (2, H, tiles, 64, 8), v (2, H, steps, 2, 64, 8) bf16."""

    def __init__(self, m, c, device, heads=8):
        self.m, self.c, self.heads = int(m), int(c), int(heads)
        self.tiles, self.steps = (self.m + 15) // 16, (self.m + 31) // 32
        self.k = torch.empty(2, heads, self.tiles, 64, 8, device=device, dtype=torch.bfloat16)
        self.v = torch.empty(2, heads, self.steps, 2, 64, 8, device=device, dtype=torch.bfloat16)

    def k_rows(self):
        """(2, tiles * 16, C): the planes as plain rows (tests)."""
        h, t = self.heads, self.tiles
        return self.k.view(2, h, t, 4, 16, 8).permute(0, 2, 4, 1, 3, 5).reshape(2, t * 16, h * 32)

    def v_rows(self):
        """(2, steps * 32, C): the planes as plain rows (tests)."""
        h, s = self.heads, self.steps
        # [plane][h][s][half][g][qi][t][r]  ->  key = 32 s + 16 t + 4 g + r, channel = 32 h + 16 half + qi
        return self.v.view(2, h, s, 2, 4, 16, 2, 4).permute(0, 2, 6, 4, 7, 1, 3, 5).reshape(2, s * 32, h * 32)


def chain_gemm_two_sources(src, src2, split, weight, bias, out, kv=None, keep_fp32=False):
    """One GEMM over a stacked weight (N, K) whose output columns [0, split) are computed from buf[src] and [split, N) from
    buf[src2] (split a multiple of 256): nn.MultiheadAttention's packed in-projection with q, k from x + pos and v from x.
    kv (KVPlanes, N = 768): the K and V columns are written as the attention core's split-bf16 operands INSTEAD of fp32 (out
    receives the Q columns only; keep_fp32 - a training step, whose attention backward reads fp32 rows: all columns)."""
    g, ldg = _rows(out, 'out')
    img, n, k = _image_of(weight, ALL_EXACT[0])
    op = ChainOp(kind=CHAIN_GEMM, src=src, dst=-1, res=src2, K=k, N=n, flags=CHAIN_SRC2 | (CHAIN_EXACT if ALL_EXACT[0] else 0), ld0=int(split),
                 ldg=ldg, p0=img, p1=None if bias is None else bias.data_ptr(), gout=g)
    if kv is not None:
        if n != 3 * kv.c or kv.c != 256 or kv.heads != 8:
            raise ValueError('kv planes: the packed in-projection of 256 channels, 8 heads')
        op.flags |= CHAIN_SPLIT_KV | (CHAIN_SPLIT_KV_KEEP if keep_fp32 else 0)
        op.p2, op.ld2 = kv.k.data_ptr(), kv.k[0].numel()
        op.p3, op.ld1 = kv.v.data_ptr(), kv.v[0].numel()
    return op


_STACKED = {}


def _stacked_linears(linears):
    """cat of the Linears' weights / biases, cached while none of them changes: the entry holds weak references to the weight
    tensors it was built from (an id() or an address alone can be handed to another object after a free) plus their version
    counters and the image epoch."""
    import weakref
    key = tuple(id(m) for m in linears)
    sig = tuple((m.weight.data_ptr(), m.weight._version, None if m.bias is None else (m.bias.data_ptr(), m.bias._version))
                for m in linears) + (_CHAIN_EPOCH[0],)
    hit = _STACKED.get(key)
    if hit is not None and hit[0] == sig and all(r() is m.weight for r, m in zip(hit[3], linears)):
        return hit[1], hit[2]
    with torch.no_grad():
        w = torch.cat([m.weight for m in linears], 0).contiguous()
        b = torch.cat([m.bias if m.bias is not None else m.weight.new_zeros(m.weight.shape[0]) for m in linears], 0).contiguous()
    if len(_STACKED) > 256:                               # modules come and go in long-lived processes
        _STACKED.clear()
    _STACKED[key] = (sig, w, b, [weakref.ref(m.weight) for m in linears])
    return w, b


def chain_gemm_three_outputs(src, linears, outs, stacked=None, exact=False):
    """Three nn.Linear of ONE input (buf[src]) as one GEMM over their stacked weights; column block i goes to outs[i] (M, N_i),
    dense rows.  The sums of a column do not depend on the operation it is part of: bit-identical to three chain_gemm.
    exact: fp32-class products (GD4D_CHAIN_EXACT) - one of the outputs are sampling offsets in metres, which a camera matrix turns
    into pixels before the visibility mask is decided."""
    if len(linears) != 3 or len(outs) != 3:
        raise ValueError('chain_gemm_three_outputs takes three Linears and three outputs')
    ptrs = [_rows(o, 'out') for o in outs]
    for (ptr, ld), m in zip(ptrs, linears):
        if ld != m.weight.shape[0]:
            raise ValueError('chain_gemm_three_outputs: every output must be dense, as wide as its Linear')
    if stacked is not None:                   # (WeightImage of the stacked weights, stacked bias) of an ImageSet
        img, n, k = _image_of(stacked[0], exact)
        b = stacked[1]
    else:
        w, b = _stacked_linears(linears)
        img, n, k = _image_of(w, exact)
    return ChainOp(kind=CHAIN_GEMM, src=src, dst=-1, res=-1, K=k, N=n, flags=CHAIN_SPLIT_OUT | (CHAIN_EXACT if exact else 0),
                   ldg=ptrs[0][1], ld2=ptrs[1][1], ld1=ptrs[2][1], p0=img, p1=b.data_ptr(),
                   gout=ptrs[0][0], p2=ptrs[1][0], p3=ptrs[2][0])


def chain_headgemm(agg, wsum, weight, bias=None, dst=-1, res=-1, out=None, addend=None):
    """value_proj of the per-head aggregates (cross_attn_agg_fwd's agg (..., Hh, K), wsum (..., Hh), contiguous) as a chain
    operation: v[m, n] = sum_k agg[m][h][k] W[n][k] + bias[n] wsum[m][h] (+ buf[res]) (+ addend[m, n]) -> buf[dst] and / or out.
    addend (M, N): a global tensor that is added - cross_attn_agg_coarse_fwd's pagg; excludes `out`."""
    heads, k = agg.shape[-2], agg.shape[-1]
    img, n, wk = _image_of(weight)
    if wk != k or n % heads or (n // heads) % 32 or wsum.numel() * k != agg.numel():
        raise ValueError('chain_headgemm: weight (N, K), agg (..., Hh, K), wsum (..., Hh) with (N / Hh) % 32 == 0')
    if addend is not None and (out is not None or dst < 0 or addend.shape[-1] != n):
        raise ValueError('chain_headgemm: an addend (M, N) goes with dst >= 0 and no `out`')
    g, ldg = _rows(out if addend is None else addend, 'out')
    return ChainOp(kind=CHAIN_HEADGEMM, src=-1, dst=dst, res=res, K=k, N=n, ld0=heads, ldg=ldg, p0=img,
                   flags=CHAIN_ADD_GOUT if addend is not None else 0,
                   p1=None if bias is None else bias.data_ptr(), p2=_dev(agg, 'agg', torch.float32).value,
                   p3=_dev(wsum, 'wsum', torch.float32).value, gout=g)


def chain_small_linear(src, weight, bias, dst, relu=False, inv_sigmoid=False, out=None):
    """buf[dst] = act(f(buf[src][:, :K]) W^T + b) for K <= 8; f = inverse_sigmoid with inv_sigmoid=True."""
    g, ldg = _rows(out, 'out')
    return ChainOp(kind=CHAIN_SMALL_LINEAR, src=src, dst=dst, res=-1, K=weight.shape[1], N=weight.shape[0], ldg=ldg, gout=g,
                   flags=(CHAIN_RELU if relu else 0) | (CHAIN_INV_SIGMOID if inv_sigmoid else 0), p0=weight.data_ptr(),
                   p1=None if bias is None else bias.data_ptr())


def chain_layernorm_bwd(src, x_buf, norm, dst=-1, relu=False, out=None, part=None):
    """Backward of chain_layernorm: buf[src] = gradient of the output; the forward's input = buf[x_buf], or - x_buf a tensor -
    its rows in global memory (no LOAD operation needed); dx -> buf[dst] (may be src) and / or out; part: (ceil(M / 16), 2, N)
    fp32 partial dgamma / dbeta (layernorm_bwd_reduce_group adds them)."""
    g, ldg = _rows(out, 'out')
    p3, ld3 = (None, 0) if isinstance(x_buf, int) else _rows(x_buf, 'x')
    return ChainOp(kind=CHAIN_LN_BWD, src=src, dst=dst, res=x_buf if isinstance(x_buf, int) else -1, N=norm.weight.shape[0],
                   eps=float(norm.eps), flags=CHAIN_RELU if relu else 0, ldg=ldg, ld1=ld3, p0=norm.weight.data_ptr(),
                   p1=norm.bias.data_ptr(), p2=None if part is None else _dev(part, 'part', torch.float32).value, gout=g, p3=p3)


def chain_layernorm(src, norm, dst=-1, relu=False, out=None, dst2=-1, add=None):
    """LayerNorm of buf[src] -> buf[dst] and / or out; with dst2 and add: also buf[dst2] = result + add[m, :] (the ADD
    operation that would follow, e.g. x + query_pos for the next projection)."""
    g, ldg = _rows(out, 'out')
    p2, ld2 = _rows(add, 'add')
    if (p2 is None) != (dst2 < 0):
        raise ValueError('chain_layernorm: dst2 and add go together')
    return ChainOp(kind=CHAIN_LAYERNORM, src=src, dst=dst, res=dst2, N=norm.weight.shape[0], eps=float(norm.eps),
                   flags=CHAIN_RELU if relu else 0, ldg=ldg, ld2=ld2, p0=norm.weight.data_ptr(), p1=norm.bias.data_ptr(),
                   p2=p2, gout=g)


def chain_add(dst, src, n, res=-1, add=None, out=None):
    """buf[dst] = buf[src] (+ buf[res]) (+ add[m, :]); out: the sum is also stored there."""
    p2, ld2 = _rows(add, 'add')
    g, ldg = _rows(out, 'out')
    return ChainOp(kind=CHAIN_ADD, src=src, dst=dst, res=res, N=n, ld2=ld2, p2=p2, ldg=ldg, gout=g)


def chain_refine(src, ref, out, dst=-1):
    """Reference-point refinement of buf[src] (the reg branch's output) and `ref` into `out`; dst >= 0 also parks the
    refined points in buf[dst][:, 0:3]."""
    return ChainOp(kind=CHAIN_REFINE, src=src, dst=dst, res=-1, p0=ref.data_ptr(), gout=out.data_ptr())


def chain_signal(flags):
    """Two-program launches: publish this program's global outputs of its 16 rows to the other program (flags: int32 tensor
    of >= ceil(M / 16) zeros, one per row block)."""
    return ChainOp(kind=CHAIN_SIGNAL, src=-1, dst=-1, res=-1, gout=_dev(flags, 'flags', torch.int32).value)


_HANDOFF = {}       # device index -> {'word': int32[1] on the device, 'pinned': int32[1] host, 'event': Event or None, 'placement': bool}


def _handoff_state(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _HANDOFF.get(idx)
    if st is None:
        dev = torch.device('cuda', idx)
        st = _HANDOFF[idx] = {'word': None, 'dev': dev, 'pinned': None, 'event': None, 'placement': None}
    return st


def handoff_error_word(device):
    """The device's sticky error word of the SIGNAL / WAIT hand-offs between chain programs: a WAIT that gives up (~0.2 s
    unanswered) adds 1 to it and POISONS the rows it hands on (NaN), so a time-out is a wrong answer nobody can mistake for a right
    one, and check_handoff() / poll_handoff() turn it into an exception."""
    st = _handoff_state(device)
    if st['word'] is None:
        st['word'] = torch.zeros(1, device=st['dev'], dtype=torch.int32)
    return st['word']


_HANDOFF_FLAGS = {}       # (device index, request slot key, rows, cols) -> int32 (rows, cols), zero between requests


def handoff_flags(device, rows, cols, slot_key):
    """The SIGNAL / WAIT flags of one request: (rows, cols) int32, zero when a request starts - the WAITing program takes every flag
    down again after it has seen it - so ONE persistent buffer per (device, request slot) serves every request and a replayed
    hipGraph holds no fill.  slot_key: functional.slot_key(device) - requests in flight on different streams get their own."""
    device = torch.device(device)
    key = (device.index, slot_key, int(rows), int(cols))
    buf = _HANDOFF_FLAGS.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(rows, cols, device=device, dtype=torch.int32)       # (first use inside a capture: this graph's own, filled per replay)
        buf = _HANDOFF_FLAGS[key] = torch.zeros(rows, cols, device=device, dtype=torch.int32)
    return buf


def check_handoff(device=None):
    """Blocking: raises Gd4dError if a hand-off has timed out on `device` (default: every device used so far) since the last check.
    Call it where a host sync exists anyway - after a request's graph replay (or every N replays), at the end of a step."""
    sts = list(_HANDOFF.items()) if device is None else [(None, _handoff_state(device))]
    for _, st in sts:
        if st['word'] is None:
            continue
        n = int(st['word'].item())
        st['event'] = None
        if n:
            st['word'].zero_()
            for buf in _HANDOFF_FLAGS.values():          # a late SIGNAL may have raised a flag its WAIT no longer took down
                buf.zero_()
            # ... and the device builds its steps WITHOUT hand-offs from here on (handoff_enabled): the probe that allowed them covers
            # one launch shape on one stream at start-up - not a CU-masked stream, a later partition-mode change or the chain's own
            # resource footprint - and a time-out is the evidence that it did not hold.  (Graphs captured earlier keep theirs.)
            st['placement'] = False
            raise _lib.Gd4dError(f'{n} SIGNAL / WAIT hand-off(s) between chain programs timed out: the affected rows are NaN. '
                                 'GD4D_POS_ENCODER=dual / GD4D_TRAIN_REG_BESIDE=0 run the same step without hand-offs.')


def poll_handoff(device):
    """Non-blocking form for eager launches (not inside a graph capture): looks at the copy of the error word an earlier call
    requested, if it has arrived, and requests the next one - a time-out surfaces one call later at the latest, without a sync."""
    if torch.cuda.is_current_stream_capturing():
        return
    st = _handoff_state(device)
    if st['word'] is None:
        return
    if st['event'] is not None and st['event'].query():
        st['event'] = None
        if int(st['pinned'][0]) != 0:
            check_handoff(device)
    if st['event'] is None:
        if st['pinned'] is None:
            st['pinned'] = torch.zeros(1, dtype=torch.int32).pin_memory()
        st['pinned'].copy_(st['word'], non_blocking=True)
        st['event'] = torch.cuda.Event()
        st['event'].record()


def handoff_placement_ok(device):
    """One-time self-test per device: the hand-offs publish their rows to the XCD's L2 only, so they need workgroup j and workgroup
    j + 8 k of a launch on the SAME XCD (what the dispatcher does today, not what any specification promises).
    gd4d_xcd_placement_probe records every workgroup's XCC id; False if the pattern does not hold (or cannot be probed because
    the first use falls inside a graph capture) - the callers then take the schedules without hand-offs."""
    st = _handoff_state(device)
    if st['placement'] is None:
        if torch.cuda.is_current_stream_capturing():
            import warnings
            warnings.warn('graph-detr4d_amd: first use inside a graph capture - the XCD placement self-test cannot run, the step is '
                          'built without SIGNAL / WAIT hand-offs (run one eager forward before capturing to get them)')
            return False
        lib = _lib.load()
        idx = torch.device(device).index
        with torch.cuda.device(idx if idx is not None else torch.cuda.current_device()):
            blocks = 2048
            out = torch.full((blocks,), -1, device=st['dev'], dtype=torch.int32)
            _lib.check(lib.gd4d_xcd_placement_probe(_dev(out, 'out', torch.int32), blocks, _stream()), 'gd4d_xcd_placement_probe')
            ids = out.cpu()
        # (the probe's own shape: 2048 workgroups of one wave on the current stream; a time-out later on turns the hand-offs off
        #  for the device, check_handoff)
        st['placement'] = bool((ids >= 0).all() and (ids == ids[:8].repeat(blocks // 8)).all())
        if not st['placement']:
            import warnings
            warnings.warn('graph-detr4d_amd: workgroups j and j + 8 k do not share an XCD on this device - chain programs run '
                          'without SIGNAL / WAIT hand-offs')
    return st['placement']


def handoff_enabled(device, env):
    """Whether a step may use hand-offs: the switch `env` (GD4D_POS_ENCODER=dual / GD4D_TRAIN_REG_BESIDE=0 turn them off) and the
    placement self-test."""
    import os
    if os.environ.get(env, '1') in ('0', 'dual'):
        return False
    return handoff_placement_ok(device)


def chain_wait(flags, errors=None):
    """Two-program launches: hold this program until the other program's workgroup of the same 16 rows has signalled on
    `flags`; errors: int32 tensor (1 element) that counts waits that gave up (handoff_error_word()).  A WAIT that gives up
    poisons what the program LOADs afterwards (NaN)."""
    return ChainOp(kind=CHAIN_WAIT, src=-1, dst=-1, res=-1, p0=_dev(flags, 'flags', torch.int32).value,
                   gout=None if errors is None else _dev(errors, 'errors', torch.int32).value)


class ChainGuest(ctypes.Structure):
    """gd4d_chain_guest (include/gd4d.h): value_proj of one decoder layer over a few pyramid levels, run by guest workgroups of a
    row-chain launch."""
    _fields_ = [('feats', ctypes.c_void_p * 8), ('level_hw', ctypes.c_int32 * 16), ('L', ctypes.c_int32), ('R', ctypes.c_int32),
                ('chlast', ctypes.c_int32), ('workgroups', ctypes.c_int32), ('image', ctypes.c_void_p), ('out', ctypes.c_void_p)]


_VP_IMAGES = {}


def value_proj_image(weight, bias=None):
    """gd4d_value_proj_image of a layer's value_proj (256, 256) weight and bias: the split-bf16 fragment image the guests of
    row_chain_fwd(..., guest=) stream through LDS.  Cached like chain_weight_image (address, version counters of weight AND bias;
    invalidate_chain_images() forgets these too)."""
    import weakref
    if not weight.is_cuda or weight.dtype != torch.float32 or tuple(weight.shape) != (256, 256) or not weight.is_contiguous():
        raise ValueError('value_proj_image: a contiguous (256, 256) float32 GPU weight')
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != 256 or not bias.is_contiguous() or bias.device != weight.device):
        raise ValueError('value_proj_image: bias (256) float32 on the weight\'s GPU')
    wb = weight._base if weight._base is not None else weight
    bb = None if bias is None else (bias._base if bias._base is not None else bias)
    key = (weight.data_ptr(), None if bias is None else bias.data_ptr())
    hit = _VP_IMAGES.get(key)
    if hit is not None and hit[0]() is wb and hit[1] == wb._version and hit[4] == _CHAIN_EPOCH[0] and \
            (bb is None or (hit[2]() is bb and hit[3] == bb._version)):
        return hit[5]
    lib = _lib.load()
    img = torch.empty(int(lib.gd4d_value_proj_image_bytes()), device=weight.device, dtype=torch.uint8)
    with torch.cuda.device(weight.device):
        code = lib.gd4d_value_proj_image(_dev(weight, 'weight', torch.float32), None if bias is None else _dev(bias, 'bias', torch.float32),
                                         _dev(img, 'image'), _stream())
    _lib.check(code, 'gd4d_value_proj_image')
    _VP_IMAGES[key] = (weakref.ref(wb), wb._version, None if bb is None else weakref.ref(bb), None if bb is None else bb._version,
                       _CHAIN_EPOCH[0], img)
    return img


def chain_guest(levels, image, out, workgroups=0):
    """A ChainGuest: project `levels` - (R, 256, H, W) / (B, N, 256, H, W) fp32, all contiguous (NCHW) or all stored channels-last -
    with the value_proj whose `image` this is into out (R, sum H W, 256) fp32."""
    chlast = [PyramidView.is_channels_last_level(t) and not t.is_contiguous() for t in levels]
    if any(chlast) != all(chlast) or (not any(chlast) and not all(t.is_contiguous() for t in levels)):
        raise ValueError('chain_guest: the levels must be all NCHW-contiguous or all channels-last')
    if any(t.dtype != torch.float32 or not t.is_cuda or t.shape[-3] != 256 for t in levels) or len(levels) > 4:
        raise ValueError('chain_guest: up to 4 float32 GPU levels of 256 channels')
    r = levels[0].numel() // (256 * levels[0].shape[-1] * levels[0].shape[-2])
    s = sum(t.shape[-1] * t.shape[-2] for t in levels)
    if tuple(out.shape) != (r, s, 256) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != levels[0].device:
        raise ValueError(f'chain_guest: out must be ({r}, {s}, 256) float32, contiguous')
    g = ChainGuest()
    for i, t in enumerate(levels):
        g.feats[i] = t.data_ptr()
        g.level_hw[2 * i], g.level_hw[2 * i + 1] = int(t.shape[-2]), int(t.shape[-1])
    g.L, g.R, g.chlast, g.workgroups = len(levels), r, int(all(chlast)), int(workgroups)
    g.image, g.out = image.data_ptr(), out.data_ptr()
    g._keep = (levels, image, out)
    return g


def value_proj_guest_fwd(guest, max_cus=0):
    """gd4d_value_proj_guest_fwd: a ChainGuest's job as a launch of its own."""
    code = _lib.load().gd4d_value_proj_guest_fwd(ctypes.byref(guest), int(max_cus), _stream())
    _lib.check(code, 'gd4d_value_proj_guest_fwd')


def _chain_fills(lib, a, na, b, nb, m, fills):
    """gd4d_row_chain_fill_fwd: the chain launch carries `fills` (PyramidGrad.take_fills) as guest workgroups."""
    jobs, start, records, fb, fn, fhh = fills
    arr = (FillJob * len(jobs))(*[FillJob(_dev(pl.buf, 'plan', torch.uint8).value, _dev(sl, 'slots').value,
                                           None if pl.order is None else _order_ptr(pl.order, fb * pl.q).value, int(base), int(pl.q))
                                   for pl, sl, base in jobs])
    code = lib.gd4d_row_chain_fill_fwd(a, na, b, nb, int(m), arr, len(jobs), _dev(start, 'start', torch.int32), _dev(records, 'records'),
                                       fb, fn, fhh, int(jobs[0][0].points), 0, _stream())
    _lib.check(code, 'gd4d_row_chain_fill_fwd')


def row_chain_fwd(program, m, guest=None, fills=None):
    """gd4d_row_chain_fwd: run the list of ChainOp over `m` rows in one launch (the tensors the operations point to must
    stay alive until the stream has run it - the callers keep them in locals / return them).  guest (ChainGuest): the launch
    also carries that value_proj job on the compute units the chain leaves idle (gd4d_row_chain_guest_fwd).  fills (training;
    PyramidGrad.take_fills): it carries those record fills of the pyramid gradient (gd4d_row_chain_fill_fwd)."""
    lib = _lib.load()
    arr = (ChainOp * len(program))(*program)
    if fills is not None:
        return _chain_fills(lib, arr, len(program), None, 0, m, fills)
    if guest is not None:
        code = lib.gd4d_row_chain_guest_fwd(arr, len(program), None, 0, int(m), ctypes.byref(guest), _stream())
        _lib.check(code, 'gd4d_row_chain_guest_fwd')
        return
    code = lib.gd4d_row_chain_fwd(arr, len(program), int(m), _stream())
    _lib.check(code, 'gd4d_row_chain_fwd')


def row_chain2_fwd(program_a, program_b, m, guest=None, fills=None):
    """gd4d_row_chain2_fwd: two independent programs over the same `m` rows in one launch (each on its own workgroups);
    guest / fills as row_chain_fwd."""
    lib = _lib.load()
    a = (ChainOp * len(program_a))(*program_a)
    b = (ChainOp * len(program_b))(*program_b)
    if fills is not None:
        return _chain_fills(lib, a, len(program_a), b, len(program_b), m, fills)
    if guest is not None:
        code = lib.gd4d_row_chain_guest_fwd(a, len(program_a), b, len(program_b), int(m), ctypes.byref(guest), _stream())
        _lib.check(code, 'gd4d_row_chain_guest_fwd')
        return
    code = lib.gd4d_row_chain2_fwd(a, len(program_a), b, len(program_b), int(m), _stream())
    _lib.check(code, 'gd4d_row_chain2_fwd')


def mha_core_presplit_fwd(q, kv, num_heads, attn_mask=None, want_lse=False, dropout_p=0., seed=None):
    """gd4d_mha_core_presplit_fwd: the self-attention core (batch 1) on K / V planes a chain GEMM wrote (KVPlanes).
    q (M, 1, C); attn_mask, want_lse, dropout_p / seed as for mha_core_fwd.  Returns (M, 1, C) [, lse]; without dropout
    bit-identical to mha_core_fwd on the fp32 rows."""
    lib = _lib.load()
    lq, b, c = q.shape
    _, _, _, _, _, _, kind, mptr, keep = _mha_args(q, q, q, num_heads, attn_mask)
    if b != 1 or lq != kv.m or c != kv.c or num_heads != kv.heads:
        raise ValueError('mha_core_presplit_fwd: batch 1, the planes of these rows')
    if not q.is_cuda or q.dtype != torch.float32 or q.stride(2) != 1:
        raise _lib.Gd4dError('q must be a float32 GPU tensor with unit channel stride')
    d = c // num_heads
    out = torch.empty(lq, 1, c, device=q.device, dtype=torch.float32)
    lse = torch.empty(lq, 1, num_heads, device=q.device, dtype=torch.float32) if want_lse else None
    sptr = _mha_seed(dropout_p, seed, q.device)
    code = lib.gd4d_mha_core_presplit_fwd(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(kv.k.data_ptr()),
                                          ctypes.c_void_p(kv.v.data_ptr()), _dev(out, 'out'), lq, num_heads, d, q.stride(0), c,
                                          kv.k[0].numel(), kv.v[0].numel(), mptr, kind, 1.0 / (d ** 0.5),
                                          None if lse is None else _dev(lse, 'lse'), float(dropout_p), sptr, _stream())
    _lib.check(code, 'gd4d_mha_core_presplit_fwd')
    return (out, lse) if want_lse else out


def _first_tensor(args):
    for a in args:
        if torch.is_tensor(a):
            return a
        if isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]):
            return a[0]
    return None


def _on_tensor_device(fn):
    """Run `fn` with the device of its first tensor argument current: _stream() then hands the C ABI that device's
    current stream and the library's per-device state (CU count, LDS attributes) refers to the right GPU, also when
    one process drives several GPUs or the tensors live on a non-current device."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        t = _first_tensor(args)
        if t is None or not t.is_cuda or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapped


_HOST_ONLY = {'linear_sum_assignment_batch', 'cross_attn_plan_bytes', 'invalidate_chain_images', 'chain_load', 'chain_gemm', 'chain_small_linear', 'chain_layernorm',
              'chain_add', 'chain_refine', 'row_chain_fwd', 'chain_weight_image', 'chain_layernorm_bwd'}
for _name, _fn in list(globals().items()):
    if inspect.isfunction(_fn) and _fn.__module__ == __name__ and not _name.startswith('_') and _name not in _HOST_ONLY:
        globals()[_name] = _on_tensor_device(_fn)
del _name, _fn
