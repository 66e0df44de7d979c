"""Seeded synthetic inputs for the decoder hot path (SURVEY.md §8d recipe).

A nuScenes-like six-camera rig replicated per temporal frame ("temporal = extra cameras",
reference: projects/mmdet3d_plugin/datasets/pipelines/loading.py:120-183), FPN-shaped feature
pyramids, queries and reference points.  Pure numpy/torch on CPU; callers move tensors to the
device.  Used by tests, bench.py and the golden-vector generator so that all three see the same
distributions.
"""
import math

import numpy as np
import torch

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]          # detr4d configs, `point_cloud_range`
IMG_SHAPE = (900, 1600, 3)                                 # un-padded nuScenes image
# 900x1600 padded to 928x1600, FPN strides 8/16/32/64 (start_level=1)
R50_LEVELS = [(116, 200), (58, 100), (29, 50), (15, 25)]
# VoVNet-99 config: start_level=0, strides 4..32
VOV_LEVELS = [(232, 400), (116, 200), (58, 100), (29, 50)]

_YAWS_DEG = (0.0, -55.0, 55.0, 180.0, 110.0, -110.0)       # front, front-right, front-left, back, back-left, back-right
_FOCALS = (1266.0, 1266.0, 1266.0, 809.0, 1266.0, 1266.0)


def camera_rig(num_frames=1, img_hw=(900, 1600), cam_radius=1.5, cam_height=1.5,
               frame_shift=0.5, dtype=np.float32):
    """lidar2img (6*num_frames, 4, 4): K @ [R|t], rows act on column vector [x, y, z, 1]^T.

    Lidar frame: x forward, y left, z up.  Camera frame: x right, y down, z forward.
    Past frame t sees the ego displaced by -frame_shift*t metres along x.
    """
    h, w = img_hw
    cx, cy = w / 2.0, h / 2.0
    mats = []
    for t in range(num_frames):
        for yaw_deg, f in zip(_YAWS_DEG, _FOCALS):
            yaw = math.radians(yaw_deg)
            fwd = np.array([math.cos(yaw), math.sin(yaw), 0.0])
            right = np.array([math.sin(yaw), -math.cos(yaw), 0.0])
            down = np.array([0.0, 0.0, -1.0])
            rot = np.stack([right, down, fwd])                      # lidar -> camera
            centre = cam_radius * fwd + np.array([-frame_shift * t, 0.0, cam_height])
            trans = -rot @ centre
            k = np.array([[f * w / 1600.0, 0.0, cx], [0.0, f * w / 1600.0, cy], [0.0, 0.0, 1.0]])
            m = np.eye(4)
            m[:3, :3] = k @ rot
            m[:3, 3] = k @ trans
            mats.append(m)
    return np.stack(mats).astype(dtype)


def make_img_metas(lidar2img, img_shape=IMG_SHAPE, batch=1, pad_shape=None):
    """The `img_metas` contract of the attention modules; `pad_shape` (read by the head's feature
    position embedding) defaults to the image height rounded up to a multiple of 32 (900 -> 928, config ...ceph.py)."""
    n = lidar2img.shape[0]
    if pad_shape is None:
        pad_shape = (-(-img_shape[0] // 32) * 32, -(-img_shape[1] // 32) * 32, 3)
    return [dict(lidar2img=[lidar2img[i] for i in range(n)], img_shape=[tuple(img_shape)] * n,
                 pad_shape=[tuple(pad_shape)] * n)
            for _ in range(batch)]


def feature_pyramid(num_cams, levels=R50_LEVELS, channels=256, batch=1, seed=0,
                    dtype=torch.float32, device='cpu'):
    """List of L tensors (B, N, C, H_l, W_l) ~ N(0, 1)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    feats = []
    for (h, w) in levels:
        t = torch.randn(batch, num_cams, channels, h, w, generator=g, dtype=torch.float32)
        feats.append(t.to(device=device, dtype=dtype))
    return feats


def queries(num_query=900, channels=256, batch=1, seed=0, device='cpu'):
    """(query, query_pos) seq-first (Q, B, C) ~ N(0,1); reference_points (B, Q, 3) ~ U(0,1)."""
    g = torch.Generator(device='cpu').manual_seed(seed + 7)
    q = torch.randn(num_query, batch, channels, generator=g)
    qp = torch.randn(num_query, batch, channels, generator=g)
    ref = torch.rand(batch, num_query, 3, generator=g)
    return q.to(device), qp.to(device), ref.to(device)


def randomise_cross_attn_(module, seed=0):
    """Make a freshly-initialised Deform3DCrossAttn non-degenerate (zero-init logits otherwise
    give uniform softmax and sigmoid(0) camera weights): SURVEY.md §8d."""
    g = torch.Generator(device='cpu').manual_seed(seed + 13)

    def _n(p, std):
        with torch.no_grad():                        # (not p.data.copy_: that would not bump the version the weight-image cache watches)
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device, p.dtype))
    _n(module.attention_weights.weight, 0.05)
    if hasattr(module, 'cam_attention_weights'):
        _n(module.cam_attention_weights.weight, 0.05)
        _n(module.cam_attention_weights.bias, 0.5)
    if hasattr(module, 'deform_sampling_offsets'):
        _n(module.deform_sampling_offsets.weight, 0.01)
    _n(module.attention_weights.bias, 0.5)
    return module


def randomise_all_(module, seed=0, std=0.05):
    """Randomise every parameter (weights ~ N(0,std), LN gains ~ 1+N(0,std)) - used for fixtures."""
    g = torch.Generator(device='cpu').manual_seed(seed + 29)
    for name, p in module.named_parameters():
        r = torch.randn(p.shape, generator=g) * std
        leaf = name.rsplit('.', 1)[-1]
        if p.dim() == 1 and leaf == 'weight':              # LayerNorm gain
            r = r + 1.0
        if 'deform_sampling_offsets.bias' in name:         # keep the metre-scale head directions
            r = p.detach().cpu() + r
        with torch.no_grad():
            p.copy_(r.to(p.device, p.dtype))
    return module
