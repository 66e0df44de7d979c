"""Feature position embedding (HIP kernels + library 1x1 convs) against the reference-generated fixture and the oracle.
GPU only."""
import pytest
import torch

from golden_io import Golden
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _module(g):
    from graph_detr4d_amd import FeaturePositionEmbedding
    m = g.meta
    mod = FeaturePositionEmbedding(embed_dims=256, depth_num=m['depth_num'], depth_start=m['depth_start'],
                                   pc_range=m['pc_range'])
    own = ('position_encoder.', 'adapt_pos3d.', 'fpe.')
    mod.load_state_dict({k: v for k, v in g.state().items() if k.startswith(own)}, strict=True)   # the head's names
    return mod.cuda().eval()


def _metas(g):
    m = g.meta
    return [dict(lidar2img=[g.arrays['lidar2img'][i] for i in range(m['num_cams'])],
                 img_shape=[tuple(s) for s in m['img_shapes']], pad_shape=[tuple(m['pad_shape'])] * m['num_cams'])]


def test_frustum_input_and_mask_match_reference_pieces():
    from graph_detr4d_amd import ops
    import numpy as np
    g = Golden('head_pe')
    m = g.meta
    l2i = g.arrays['lidar2img'][None].astype(np.float64)
    i2l = torch.from_numpy(np.linalg.inv(l2i)).float().view(-1, 4, 4).cuda()
    for lvl, (h, w) in enumerate(m['levels']):
        x, outside = ops.frustum_pe_input_fwd(i2l, (h, w), m['pad_shape'][:2], m['depth_num'], m['depth_start'],
                                              m['pc_range'])
        c3, ref_out = O.frustum_points(l2i, (h, w), m['pad_shape'][:2], m['depth_num'], m['depth_start'], m['pc_range'])
        ref_x = O.inverse_sigmoid(c3.permute(0, 1, 4, 5, 3, 2).contiguous().view(-1, 3 * m['depth_num'], h, w))
        assert torch.equal(outside.cpu(), ref_out.permute(0, 1, 3, 2).reshape(-1, h, w))
        # inverse_sigmoid amplifies fp32 rounding of the coordinate near the clamps; compare where it is well conditioned
        d = (x.cpu() - ref_x).abs()
        assert d[ref_x.abs() < 8].max().item() < 2e-4
        assert (d > 1e-2).float().mean().item() < 1e-3
        mask = g.t(f'mask{lvl}').bool() | outside.view(1, -1, h, w).cpu()
        assert torch.equal(mask, g.t(f'coords_mask{lvl}').bool())


def test_sine_embedding_matches_reference():
    g = Golden('head_pe')
    mod = _module(g)
    for lvl in range(len(g.meta['levels'])):
        got = mod.sine_embedding(g.t(f'mask{lvl}').bool().cuda())
        torch.testing.assert_close(got.cpu(), g.t(f'sine{lvl}'), rtol=0, atol=2e-6)


def test_feature_position_embedding_matches_reference():
    g = Golden('head_pe')
    mod = _module(g)
    feats = [f.cuda() for f in g.feats()]
    with torch.no_grad():                                 # (with autograd on the module takes its training path)
        outs = mod(feats, _metas(g))                      # default: dense part on gd4d_gemm_bf16x3_fwd
        for lvl, o in enumerate(outs):
            torch.testing.assert_close(o.cpu(), g.t(f'out{lvl}'), rtol=2e-4, atol=2e-4)
        import os
        os.environ['GD4D_HEAD_PE'] = 'conv'               # library 1x1 convolutions
        try:
            for lvl, o in enumerate(mod(feats, _metas(g))):
                torch.testing.assert_close(o.cpu(), g.t(f'out{lvl}'), rtol=2e-4, atol=2e-4)
        finally:
            os.environ.pop('GD4D_HEAD_PE')
        again = mod(feats, _metas(g))                     # second call: sine branch from the cache
    assert mod._sine_cache is not None
    for a, b in zip(outs, again):
        assert torch.equal(a, b)
    from graph_detr4d_amd._lib import Gd4dError
    with pytest.raises(Gd4dError):
        mod.cpu()([f.cpu() for f in feats], _metas(g))


def test_feature_position_embedding_trains():
    """With autograd on, the stage equals the inference path and its gradients - to the feature maps (the backbone's),
    the two position MLPs and the SE gate - equal autograd of the oracle (a restatement of detr3d_head_pe.py:525-557)."""
    import numpy as np
    g = Golden('head_pe')
    m = g.meta
    mod = _module(g)
    with torch.no_grad():
        want_out = mod([f.cuda() for f in g.feats()], _metas(g))
    feats = [f.cuda().requires_grad_() for f in g.feats()]
    outs = mod(feats, _metas(g))
    for o, w in zip(outs, want_out):
        torch.testing.assert_close(o, w, rtol=2e-4, atol=2e-4)
    gen = torch.Generator().manual_seed(3)
    probes = [torch.randn(o.shape, generator=gen) for o in outs]
    sum((o * p.cuda()).sum() for o, p in zip(outs, probes)).backward()
    # oracle, CPU autograd
    params = {k: v.clone().requires_grad_() for k, v in g.state().items() if k.startswith(('position_encoder.', 'adapt_pos3d.', 'fpe.'))}
    feats_c = [f.clone().requires_grad_() for f in g.feats()]
    l2i = torch.from_numpy(g.arrays['lidar2img'][None].astype(np.float64))
    o_outs, _ = O.feature_position_embedding(params, feats_c, l2i, [[tuple(s) for s in m['img_shapes']]], tuple(m['pad_shape']),
                                             m['depth_num'], m['depth_start'], m['pc_range'])
    sum((o * p).sum() for o, p in zip(o_outs, probes)).backward()
    for fg, fc in zip(feats, feats_c):
        torch.testing.assert_close(fg.grad.cpu(), fc.grad, rtol=1e-3, atol=1e-4)
    for name, prm in mod.named_parameters():
        want = params[name].grad
        assert prm.grad is not None, name
        tol = 2e-3 * max(1.0, want.abs().max().item())
        assert (prm.grad.cpu() - want).abs().max().item() < tol, name


def test_se_fuse_equals_torch():
    from graph_detr4d_amd import ops
    torch.manual_seed(0)
    f, g_, pe, s = (torch.randn(3, 256, 9, 4).cuda() for _ in range(4))
    got = ops.se_fuse_fwd(f, g_, pe, s)
    torch.testing.assert_close(got, f + (pe * torch.sigmoid(g_) + s), rtol=1e-6, atol=1e-6)


def test_head_outputs_match_reference_forward():
    """functional.head_outputs (cls / reg branches on the HIP linears + gd4d_box_head_fwd) against what the reference's
    Detr3DHeadPE.forward returned for the same decoder outputs."""
    import torch.nn as nn
    from graph_detr4d_amd import functional as Fn
    g = Golden('head_pe')
    m, sd = g.meta, g.state()
    nl = m['num_layers']

    def cls():
        return nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                             nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True), nn.Linear(256, 10))

    def reg():
        return nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 10))
    cls_b, reg_b = nn.ModuleList(cls() for _ in range(nl)), nn.ModuleList(reg() for _ in range(nl))
    cls_b.load_state_dict({k[len('cls_branches.'):]: v for k, v in sd.items() if k.startswith('cls_branches.')})
    reg_b.load_state_dict({k[len('reg_branches.'):]: v for k, v in sd.items() if k.startswith('reg_branches.')})
    cls_b.cuda().eval(), reg_b.cuda().eval()
    with torch.no_grad():
        got = Fn.head_outputs(g.t('hs').cuda(), g.t('init_reference').cuda(), g.t('inter_references').cuda(), cls_b,
                              reg_b, m['pc_range'])
    torch.testing.assert_close(got['all_cls_scores'].cpu(), g.t('all_cls_scores'), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got['all_bbox_preds'].cpu(), g.t('all_bbox_preds'), rtol=1e-4, atol=1e-4)


def test_with_detach_cuts_the_gradient_of_level_0_past_frames():
    """detr3d_head_pe.py:512-516 (`with_detach`, default True): level 0's cameras after the first six reach the stage
    detached - same outputs, no gradient to the backbone through them; every other (level, camera) keeps its gradient.
    with_detach=False restores it."""
    from graph_detr4d_amd import FeaturePositionEmbedding, synthetic
    img_hw, levels, n = (64, 112), [(8, 14), (4, 7)], 12
    rig = synthetic.camera_rig(2, img_hw)
    metas = synthetic.make_img_metas(rig, img_shape=(img_hw[0], img_hw[1], 3), pad_shape=(img_hw[0], img_hw[1], 3))
    gen = torch.Generator().manual_seed(5)
    base = [torch.randn(1, n, 256, h, w, generator=gen) for h, w in levels]
    probes = [torch.randn(1, n, 256, h, w, generator=gen).cuda() for h, w in levels]
    res = {}
    for detach in (True, False):
        mod = FeaturePositionEmbedding(pc_range=synthetic.PC_RANGE, with_detach=detach)
        synthetic.randomise_all_(mod, seed=9, std=0.04)
        mod = mod.cuda().eval()
        feats = [f.cuda().requires_grad_() for f in base]
        outs = mod(feats, metas)
        sum((o * p).sum() for o, p in zip(outs, probes)).backward()
        res[detach] = ([o.detach() for o in outs], [f.grad for f in feats])
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)                                        # values do not change
    g_det, g_full = res[True][1], res[False][1]
    assert float(g_det[0][:, 6:].abs().max()) == 0.0 and float(g_full[0][:, 6:].abs().max()) > 0
    assert torch.equal(g_det[0][:, :6], g_full[0][:, :6]) and torch.equal(g_det[1], g_full[1])
    assert FeaturePositionEmbedding(pc_range=synthetic.PC_RANGE).with_detach is True      # the head's default (:326)
