"""ORACLE - test infrastructure.  ctypes loader for oracle/_build/libgd4d_oracle.so (plain C)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get('GD4D_ORACLE_SO') or os.path.join(_DIR, '_build', 'libgd4d_oracle.so')   # (env: the sanitizer build, tools/sanitize_cpu.sh)
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _DIR])
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def cross_attn_fwd(value, level_hw, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                   img_h, img_w):
    """numpy in / numpy out; same argument meaning as gd4d_cross_attn_fwd.
    Returns out (B,Q,C), mask (B,N,Q,Hh,P) uint8, uv (B,N,Q,Hh,P,2)."""
    lib = load()
    value, ref, offsets = _f32(value), _f32(ref), _f32(offsets)
    attn_logits, cam_logits, lidar2img = _f32(attn_logits), _f32(cam_logits), _f32(lidar2img)
    b, q = ref.shape[:2]
    n = lidar2img.shape[1]
    hh, dh = value.shape[2], value.shape[3]
    p = offsets.shape[3]
    nl = len(level_hw)
    lv = np.asarray(level_hw, dtype=np.int32).reshape(-1)
    rng = np.asarray(pc_range, dtype=np.float64)
    out = np.empty((b, q, hh * dh), np.float32)
    mask = np.empty((b, n, q, hh, p), np.uint8)
    uv = np.empty((b, n, q, hh, p, 2), np.float32)
    rc = lib.gd4d_oracle_cross_attn_fwd(
        _p(value), _p(lv), _p(ref), _p(offsets), _p(attn_logits), _p(cam_logits), _p(lidar2img),
        _p(rng), ctypes.c_float(img_h), ctypes.c_float(img_w), _p(out), _p(mask), _p(uv),
        b, n, q, hh, dh, nl, p)
    assert rc == 0
    return out, mask, uv


def detr3d_fwd(feats, ref, attn_logits, lidar2img, pc_range, img_h, img_w, want_sampled=False):
    """feats: list of L arrays (B,N,C,H,W); attn_logits (B,Q,N,1,L).  Returns out (B,Q,C),
    mask (B,N,Q) uint8, uv (B,N,Q,2) [, sampled (B,C,Q,N,1,L)]."""
    lib = load()
    feats = [_f32(f) for f in feats]
    ref, attn_logits, lidar2img = _f32(ref), _f32(attn_logits), _f32(lidar2img)
    b, n, c = feats[0].shape[:3]
    q = ref.shape[1]
    nl = len(feats)
    lv = np.asarray([f.shape[3:] for f in feats], dtype=np.int32).reshape(-1)
    ptrs = (ctypes.c_void_p * nl)(*[f.ctypes.data for f in feats])
    rng = np.asarray(pc_range, dtype=np.float64)
    out = np.empty((b, q, c), np.float32)
    mask = np.empty((b, n, q), np.uint8)
    uv = np.empty((b, n, q, 2), np.float32)
    sampled = np.empty((b, c, q, n, 1, nl), np.float32) if want_sampled else None
    rc = lib.gd4d_oracle_detr3d_fwd(ptrs, _p(lv), _p(ref), _p(attn_logits), _p(lidar2img), _p(rng),
                                    ctypes.c_float(img_h), ctypes.c_float(img_w), _p(out), _p(mask),
                                    _p(uv), _p(sampled), b, n, q, c, nl)
    assert rc == 0
    return (out, mask, uv, sampled) if want_sampled else (out, mask, uv)
