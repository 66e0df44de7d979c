"""Pin the CPU oracle (oracle/torch_oracle.py) against golden vectors captured from the
reference itself (tools/gen_golden.py).  CPU only."""
import pytest
import torch

from golden_io import Golden, sub
from oracle import torch_oracle as O

DEFORM_CASES = ['deform_n6', 'deform_n12_depth', 'deform_n24_b2', 'deform_edge']
DETR3D_CASES = ['detr3d_n6', 'detr3d_n12_b2']


@pytest.mark.parametrize('name', DEFORM_CASES)
def test_deform3d_cross_attn_matches_reference(name):
    g = Golden(name)
    m = g.meta
    res, parts = O.deform3d_cross_attn(
        g.state(), g.t('query'), g.feats(), g.t('query_pos'), g.t('reference_points'),
        g.img_metas(), m['pc_range'], m['num_heads'], m['num_points'],
        depth_encode=m['depth_encode'], return_parts=True)
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    # visibility mask: bit exact.  golden mask is (B*N, Q, Hh, L*P) with P fastest; all levels equal
    gm = g.t('mask').view(b, n, q, 8, 4, 4)
    assert torch.equal(gm[..., 0, :], gm[..., 3, :])
    assert torch.equal(parts['mask'].to(torch.uint8), gm[..., 0, :])
    guv = g.t('uv').view(b, n, q, 8, 4, 4, 2)[..., 0, :, :]
    assert torch.equal(parts['uv'], guv)                       # same ATen ops -> bit exact here
    torch.testing.assert_close(parts['agg'], g.t('agg'), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(parts['pos'].permute(1, 0, 2), g.t('pos_feat'), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res, g.t('out'), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', DEFORM_CASES)
def test_sample_aggregate_contract(name):
    """The fused-kernel contract fed with the reference's own intermediates."""
    g = Golden(name)
    m = g.meta
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    sd = g.state()
    flat, shapes = O.flatten_pyramid(g.feats())
    val = torch.nn.functional.linear(flat, sd['value_proj.weight'], sd['value_proj.bias'])
    val = val.view(b * n, -1, 8, 32)
    l2i = torch.from_numpy(g.arrays['lidar2img']).unsqueeze(0).expand(b, -1, -1, -1)
    agg, uv, mask = O.sample_aggregate(
        val, shapes, g.t('reference_points'), g.t('offsets').view(b, q, 8, 4, 3),
        g.t('attn_logits').view(b, q, 8, 16), g.t('cam_logits'), l2i, m['pc_range'],
        m['img_shape'][0], m['img_shape'][1])
    torch.testing.assert_close(agg, g.t('agg'), rtol=1e-5, atol=1e-5)
    gm = g.t('mask').view(b, n, q, 8, 4, 4)[..., 0, :]
    assert torch.equal(mask.to(torch.uint8), gm)


def test_explicit_bilinear_equals_grid_sample():
    """oracle.bilinear_zero_pad (explicit gather) == ATen grid_sample incl. border zero padding."""
    torch.manual_seed(0)
    h, w, d = 5, 7, 8
    v = torch.randn(h, w, d)
    loc = torch.rand(200, 2) * 1.2 - 0.1
    loc[:6] = torch.tensor([[0., 0.], [1., 1.], [0.5 / w, 0.5 / h], [1 - 0.5 / w, 0.3], [1e-7, 0.5],
                            [0.999999, 0.999999]])
    ours = O.bilinear_zero_pad(v, loc[:, 0] * w - 0.5, loc[:, 1] * h - 0.5)
    ref = torch.nn.functional.grid_sample(v.permute(2, 0, 1)[None], (2 * loc - 1).view(1, -1, 1, 2),
                                          mode='bilinear', padding_mode='zeros', align_corners=False)
    torch.testing.assert_close(ours, ref[0, :, :, 0].T, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', DETR3D_CASES)
def test_detr3d_cross_atten_matches_reference(name):
    g = Golden(name)
    m = g.meta
    ref3d, sampled, mask = O.feature_sampling(g.feats(), g.t('reference_points'), m['pc_range'],
                                              g.img_metas())
    assert torch.equal(mask.to(torch.uint8), g.t('fs_mask'))
    torch.testing.assert_close(sampled, g.t('fs_sampled'), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ref3d, g.t('fs_ref3d'), rtol=0, atol=0)
    out = O.detr3d_cross_atten(g.state(), g.t('query'), g.feats(), g.t('query_pos'),
                               g.t('reference_points'), g.img_metas(), m['pc_range'],
                               m['num_points'])
    torch.testing.assert_close(out, g.t('out'), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', ['self_attn', 'self_attn_mask'])
def test_self_attention_matches_aten(name):
    g = Golden(name)
    mask = g.t('attn_mask').bool() if g.has('attn_mask') else None
    out = O.multihead_self_attn(g.state(), g.t('query'), g.t('query_pos'), g.meta['num_heads'],
                                attn_mask=mask)
    torch.testing.assert_close(out, g.t('out'), rtol=1e-5, atol=2e-5)


def test_hdetr_transformer_matches_the_reference_classes():
    """decoder_hdetr: the reference's own HDetr3DTransformer (utils/h_detr3d_transformer.py:49-175) driven with the block mask its head
    builds (dense_heads/h_detr3d_head_pe.py:299-304, handed over as [mask, None]) - the masked self-attention path pinned through
    the reference's CLASSES, not only through the self_attn_mask fixture."""
    g = Golden('decoder_hdetr')
    m = g.meta
    sd = g.state()
    layers = [sub(sd, f'decoder.layers.{i}.') for i in range(m['num_layers'])]
    reg_sd = g.state(prefix='reg.')

    def reg(i):
        p = sub(reg_sd, f'{i}.')
        F = torch.nn.functional
        return lambda x: F.linear(F.relu(F.linear(F.relu(F.linear(
            x, p['0.weight'], p['0.bias'])), p['2.weight'], p['2.bias'])), p['4.weight'], p['4.bias'])
    mask = g.t('self_attn_mask').bool()
    k = m['num_queries_one2one']
    assert mask[k:, :k].all() and mask[:k, k:].all() and not mask[:k, :k].any() and not mask[k:, k:].any()
    states, init_ref, refs = O.transformer(
        sd, layers, g.feats(), g.t('query_embed'), g.img_metas(), m['pc_range'],
        reg_branches=[reg(i) for i in range(m['num_layers'])], cross=m['cross'], num_points=m['num_points'], attn_mask=mask)
    torch.testing.assert_close(init_ref, g.t('init_reference'), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(refs, g.t('inter_references'), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(states, g.t('inter_states'), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('name', ['decoder_deform', 'decoder_detr3d'])
def test_transformer_decoder_matches_reference(name):
    g = Golden(name)
    m = g.meta
    sd = g.state()
    layers = [sub(sd, f'decoder.layers.{i}.') for i in range(m['num_layers'])]
    reg_sd = g.state(prefix='reg.')

    def reg(i):
        p = sub(reg_sd, f'{i}.')
        F = torch.nn.functional
        return lambda x: F.linear(F.relu(F.linear(F.relu(F.linear(
            x, p['0.weight'], p['0.bias'])), p['2.weight'], p['2.bias'])), p['4.weight'], p['4.bias'])
    states, init_ref, refs = O.transformer(
        sd, layers, g.feats(), g.t('query_embed'), g.img_metas(), m['pc_range'],
        reg_branches=[reg(i) for i in range(m['num_layers'])], cross=m['cross'],
        num_points=m['num_points'])
    torch.testing.assert_close(init_ref, g.t('init_reference'), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(refs, g.t('inter_references'), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(states, g.t('inter_states'), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('name', ['deform_mp_n6', 'deform_mp_n12_b2'])
def test_deform3d_cross_attn_mp_matches_reference(name):
    """Deform3DCrossAttnMP: the reference's forward (its second MSDA call completed by the refstub frame hook)."""
    g = Golden(name)
    m = g.meta
    out, parts = O.deform3d_cross_attn_mp(g.state(), g.t('query'), g.feats(), g.t('reference_points'), g.img_metas(),
                                          m['pc_range'], return_parts=True)
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    gmask = g.t('neighbor_mask').view(b, n, 8 * q, 8, 4)[..., 0]           # equal for all levels
    assert torch.equal(parts['mask_n'][..., 0], gmask.bool())
    torch.testing.assert_close(parts['blend'], g.t('blend_logits'), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(parts['mixed'], g.t('blended'), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out, g.t('out'), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('name', ['dgcnn', 'dgcnn_k8'])
def test_dgcnn_attn_matches_reference(name):
    g = Golden(name)
    out, parts = O.dgcnn_attn(g.state(), g.t('query'), g.t('query_pos'), g.meta['K'], return_parts=True)
    assert torch.equal(parts['idx1'], g.t('idx1'))
    torch.testing.assert_close(parts['f1'], g.t('f1'), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out, g.t('out'), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', ['detr3d_v2_n6', 'detr3d_v2_n12'])
def test_detr3d_cross_atten_v2_matches_reference(name):
    g = Golden(name)
    m = g.meta
    out, parts = O.detr3d_cross_atten_v2(g.state(), g.t('query'), g.feats(), g.t('query_pos'), g.t('reference_points'),
                                         g.img_metas(), m['pc_range'], return_parts=True)
    gmask = g.t('mask').view(1, m['num_query'], m['num_cams']).permute(0, 2, 1).bool()     # (1,1,Q,N,1,1) -> (B,N,Q)
    assert torch.equal(parts['mask'], gmask)
    assert 0 < gmask.float().mean().item() < 1
    torch.testing.assert_close(parts['agg'], g.t('agg'), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out, g.t('out'), rtol=1e-4, atol=1e-4)
