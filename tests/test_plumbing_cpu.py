"""configs[0] plumbing without a GPU: images -> ResNet18 + FPN stand-in -> (B, N, C, H, W) levels
(projects/mmdet3d_plugin/models/detectors/detr3d.py:39-66), and the CPU oracle of the 1-layer DETR3D decoder runs on them.
The HIP decoder on the same inputs is checked against that oracle in tests/test_configs_gpu.py."""
import torch

from config_cases import config0, oracle_params
from graph_detr4d_amd import plumbing, synthetic


def test_extractor_folds_cameras_and_unfolds_levels():
    c = config0()
    feats = c['feats']
    assert [tuple(f.shape) for f in feats] == [(1, 6, 256, 32, 32), (1, 6, 256, 16, 16), (1, 6, 256, 8, 8),
                                               (1, 6, 256, 4, 4)]
    assert all(f.dtype == torch.float32 and f.is_contiguous() for f in feats)
    assert c['metas'][0]['input_shape'] == (256, 256)                       # detr3d.py:43-46
    # folding is a pure reshape: camera n of sample b is row b*N + n of the backbone batch
    ex = c['extractor']
    img = torch.randn(2, 3, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    metas = [dict(), dict()]
    with torch.no_grad():
        both = ex(img, metas)
        one = ex(img[1], [dict()])                                           # (N, 3, H, W): a single sample
    for fb, fo in zip(both, one):
        assert fb.shape[:2] == (2, 3) and fo.shape[:2] == (1, 3)
        torch.testing.assert_close(fb[1], fo[0])
    assert ex(None, []) is None


def test_extractor_hands_fp32_to_the_decoder_whatever_the_backbone_precision():
    """@auto_fp16(apply_to=('img'), out_fp32=True) on extract_feat (detr3d.py:68-72): fp32 out."""
    class Half(torch.nn.Module):
        def forward(self, x):
            return [x.to(torch.bfloat16)[:, :, ::8, ::8], x.to(torch.bfloat16)[:, :, ::16, ::16]]
    ex = plumbing.ImageFeatureExtractor(Half())
    out = ex(torch.randn(1, 2, 3, 32, 32), [dict()])
    assert all(f.dtype == torch.float32 for f in out) and out[0].shape == (1, 2, 3, 4, 4)


def test_config0_oracle_runs_on_cpu():
    from oracle import torch_oracle as O
    c = config0()
    sd, layers = oracle_params(c['tr'])
    with torch.no_grad():
        states, init_ref, refs = O.transformer(sd, layers, c['feats'], c['query_embed'], c['metas'],
                                               synthetic.PC_RANGE, reg_branches=list(c['regs']),
                                               cross='Detr3DCrossAtten', num_points=1)
    assert states.shape == (1, 100, 1, 256) and refs.shape == (1, 1, 100, 3) and init_ref.shape == (1, 100, 3)
    assert torch.isfinite(states).all()
    # the rig sees a useful share of the 100 reference points (non-degenerate case)
    _, _, mask = O.feature_sampling(c['feats'], init_ref, synthetic.PC_RANGE, c['metas'])
    frac = mask.float().mean().item()
    assert 0.05 < frac < 0.6, frac


def test_extractor_can_hand_over_channels_last_levels():
    """channels_last=True: the same values, stored (B, N, H, W, C) - what the decoder's gather reads in place
    (ops.PyramidView.channels_last_levels) instead of re-laying the maps out (deform3d_cross_attn.py:264-276)."""
    from graph_detr4d_amd import ops
    c = config0()
    ex = plumbing.ImageFeatureExtractor(c['extractor'].img_backbone, channels_last=True)
    img = torch.randn(2, 3, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        nhwc = ex(img, [dict(), dict()])
        nchw = c['extractor'](img, [dict(), dict()])
    for a, b in zip(nhwc, nchw):
        assert a.shape == b.shape and torch.equal(a, b)
        assert ops.PyramidView.is_channels_last_level(a)
        assert a.shape[-1] * a.shape[-2] == 1 or not ops.PyramidView.is_channels_last_level(b)   # (a 1 x 1 map is both)
        assert a.permute(0, 1, 3, 4, 2).is_contiguous()                     # (B, N, H, W, C) in memory


def test_pyramid_view_addresses_every_source_the_same_way():
    """ops.PyramidView is pointers + byte strides (include/gd4d.h: address(level, row, pixel, slice, k) = level_ptr +
    row * cam_stride + pixel * pix_stride + slice * slice_stride + k * elem).  Checked here on the host, byte for byte,
    for the three sources the gather reads: the slice-planar copy, the pixel-major copy, channels-last levels in place."""
    import ctypes
    import numpy as np
    from oracle import torch_oracle as O
    from graph_detr4d_amd import ops
    torch.manual_seed(5)
    b, n = 2, 3
    levels = [(5, 7), (3, 4), (1, 2)]
    feats = [torch.randn(b, n, 256, h, w) for h, w in levels]
    flat, shapes = O.flatten_pyramid(feats)                                   # (B*N, S, 256): the reference's flatten / transpose / cat
    r, s = b * n, flat.shape[1]
    sp = flat.reshape(r, s, 8, 32).permute(2, 0, 1, 3).contiguous()           # what gd4d_pyramid_slice_planar_fwd writes
    cl = flat.reshape(r, s, 256).contiguous()                                 # what gd4d_pyramid_channels_last_fwd writes
    nhwc = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
    views = {'planar': ops.PyramidView.slice_planar(sp, shapes), 'pixel': ops.PyramidView.pixel_major(cl, shapes),
             'in place': ops.PyramidView.channels_last_levels(nhwc)}
    starts = np.cumsum([0] + [h * w for h, w in shapes])

    def read(addr):
        return ctypes.c_float.from_address(addr).value
    rng = np.random.default_rng(0)
    for name, v in views.items():
        assert v.rows == r and v.level_hw == [tuple(x) for x in shapes] and v.dtype == torch.float32
        for _ in range(200):
            lvl = int(rng.integers(len(levels)))
            row, pix = int(rng.integers(r)), int(rng.integers(shapes[lvl][0] * shapes[lvl][1]))
            sl, k = int(rng.integers(8)), int(rng.integers(32))
            addr = v.ptrs[lvl] + row * v.cam_stride[lvl] + pix * v.pix_stride + sl * v.slice_stride + 4 * k
            assert read(addr) == flat[row, starts[lvl] + pix, 32 * sl + k].item(), (name, lvl, row, pix, sl, k)
    # what is NOT accepted
    import pytest
    with pytest.raises(ValueError):
        ops.PyramidView.channels_last_levels(feats)                           # NCHW storage
    with pytest.raises(ValueError):
        ops.PyramidView.slice_planar(sp[:, :, :-1], shapes)                   # wrong pixel count
