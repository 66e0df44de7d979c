"""CPU: the oracle's restatement of Detr3DHeadPE's feature position embedding against the reference-generated
fixture (position_embeding, SELayer and SinePositionalEncoding3D are the reference's code in that fixture)."""
import torch

from golden_io import Golden
from oracle import torch_oracle as O


def test_feature_position_embedding_matches_reference():
    g = Golden('head_pe')
    m = g.meta
    feats = g.feats()
    l2i = g.arrays['lidar2img'][None]
    outs, mid = O.feature_position_embedding(g.state(), feats, l2i, [m['img_shapes']], m['pad_shape'], m['depth_num'],
                                             m['depth_start'], m['pc_range'])
    for lvl in range(len(feats)):
        assert torch.equal(mid['masks'][lvl], g.t(f'mask{lvl}').bool())
        assert torch.equal(mid['coords_masks'][lvl], g.t(f'coords_mask{lvl}').bool())
        torch.testing.assert_close(mid['sine'][lvl], g.t(f'sine{lvl}'), rtol=0, atol=0)
        torch.testing.assert_close(mid['coords_pe'][lvl], g.t(f'coords_pe{lvl}'), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(outs[lvl], g.t(f'out{lvl}'), rtol=1e-5, atol=1e-5)
    assert 0.2 < mid['coords_masks'][0].float().mean().item() < 0.8, 'fixture must exercise the range test'
    assert mid['masks'][0].any() and not mid['masks'][0].all()


def test_frustum_depths_are_the_lid_bins():
    d = O.frustum_depths(64, 1, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
    assert d.shape == (64,) and d[0].item() == 1.0
    assert bool((d[1:] > d[:-1]).all()) and d[-1].item() < 51.2


def _branch(sd, prefix, x, with_ln):
    """cls / reg branch of the head (:368-383) applied with the fixture's parameters."""
    F = torch.nn.functional
    i = 0
    for _ in range(2):
        x = F.linear(x, sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'])
        i += 1
        if with_ln:
            x = F.layer_norm(x, (x.shape[-1],), sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'])
            i += 1
        x = torch.relu(x)
        i += 1
    return F.linear(x, sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'])


def test_head_epilogue_matches_reference_forward():
    """all_cls_scores / all_bbox_preds returned by the reference's Detr3DHeadPE.forward (:568-612) for prepared decoder
    outputs: pins the oracle's box_head restatement."""
    g = Golden('head_pe')
    m, sd = g.meta, g.state()
    hs = g.t('hs').permute(0, 2, 1, 3)                     # (nl, B, Q, C), :568
    refs = [g.t('init_reference')] + list(g.t('inter_references'))[:-1]
    for lvl in range(m['num_layers']):
        cls = _branch(sd, f'cls_branches.{lvl}.', hs[lvl], True)
        box = O.box_head(_branch(sd, f'reg_branches.{lvl}.', hs[lvl], False), refs[lvl], m['pc_range'])
        torch.testing.assert_close(cls, g.t('all_cls_scores')[lvl], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(box, g.t('all_bbox_preds')[lvl], rtol=1e-5, atol=2e-5)
