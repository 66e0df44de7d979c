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
