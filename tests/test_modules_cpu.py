"""Host-side logic of the drop-in boundary that needs no GPU: registry, constructor contracts,
state-dict key names, init conventions, loud failure on CPU tensors."""
import pytest
import torch

import graph_detr4d_amd as G
from golden_io import Golden
from graph_detr4d_amd import _lib

PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def _decoder_cfg(cross, n=6, layers=2):
    return dict(type='Detr3DTransformerDecoder', num_layers=layers, return_intermediate=True,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1), cross],
                    feedforward_channels=512, ffn_dropout=0.1,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))


def test_registry_builds_reference_type_names():
    for t in ('Deform3DCrossAttn', 'Detr3DCrossAtten', 'MultiheadAttention'):
        assert t in G.ATTENTION
    m = G.build_attention(dict(type='Deform3DCrossAttn', num_cams=12, pc_range=PC, num_points=4,
                               embed_dims=256), dict(batch_first=False))
    assert isinstance(m, G.Deform3DCrossAttn) and m.num_cams == 12 and m.num_levels == 4
    with pytest.raises(KeyError):
        G.build_attention(dict(type='Detr3DCrossAttenMP'))     # dangling name in the reference too
    with pytest.raises(ValueError):
        G.Deform3DCrossAttn(embed_dims=250, num_heads=8)


@pytest.mark.parametrize('name', ['decoder_deform', 'decoder_detr3d'])
def test_state_dict_keys_match_reference(name):
    """Key names AND shapes equal those of the reference model the golden was captured from."""
    g = Golden(name)
    m = g.meta
    cross = (dict(type='Deform3DCrossAttn', num_cams=m['num_cams'], pc_range=PC, num_points=4, embed_dims=256)
             if m['cross'] == 'Deform3DCrossAttn' else
             dict(type='Detr3DCrossAtten', num_cams=m['num_cams'], pc_range=PC, num_points=1, embed_dims=256))
    tr = G.Detr3DTransformer(num_feature_levels=4, num_cams=m['num_cams'],
                             decoder=_decoder_cfg(cross, m['num_cams'], m['num_layers']))
    ours = {k: tuple(v.shape) for k, v in tr.state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in g.state().items()}
    assert ours == ref
    tr.load_state_dict(g.state(), strict=True)


def test_init_weight_matches_reference_convention():
    """deform3d_cross_attn.py:129-150: zero logits, head directions x (i+1) metres in the bias."""
    m = G.Deform3DCrossAttn(num_cams=6, pc_range=PC, num_points=4)
    assert m.attention_weights.weight.abs().sum() == 0 and m.cam_attention_weights.weight.abs().sum() == 0
    assert m.deform_sampling_offsets.weight.abs().sum() == 0
    bias = m.deform_sampling_offsets.bias.view(8, 4, 3)
    torch.testing.assert_close(bias[0], torch.tensor([[1., 0., 1.]]) * torch.arange(1, 5.)[:, None])
    torch.testing.assert_close(bias[:, 1], bias[:, 0] * 2)
    assert bias.abs().amax(-1).allclose(torch.arange(1, 5.).expand(8, 4))
    g = Golden('deform_n6')          # a reference module's untouched-by-quantisation layout
    assert set(k for k in g.state()) == set(m.state_dict().keys())


def test_forward_fails_loudly_without_gpu_and_with_residual():
    m = G.Deform3DCrossAttn(num_cams=6, pc_range=PC, num_points=4).eval()
    q = torch.zeros(5, 1, 256)
    feats = [torch.zeros(1, 6, 256, 4, 4)] * 4
    metas = [dict(lidar2img=[torch.eye(4).numpy()] * 6, img_shape=[(32, 32, 3)] * 6)]
    with torch.no_grad():
        with pytest.raises(_lib.Gd4dError):
            m(q, None, feats, None, query_pos=q, reference_points=torch.rand(1, 5, 3), img_metas=metas)
        with pytest.raises(NameError):
            m(q, None, feats, q, query_pos=q, reference_points=torch.rand(1, 5, 3), img_metas=metas)
    with pytest.raises(KeyError):
        with torch.no_grad():
            m(q, None, feats, None, query_pos=q, reference_points=torch.rand(1, 5, 3))   # img_metas mandatory


def test_chain_training_path_host_logic():
    """What graph_detr4d_amd/fused_train.py decides on the host, no GPU needed: the parameter order it hands autograd, where a
    layer's dropouts sit (mmcv's MultiheadAttention / FFN and Deform3DCrossAttn, config ...ceph.py:71-89), when the path refuses
    (CPU tensors, two dropouts in a row that are not one Bernoulli mask), and the dropout threshold / scale the chain operations
    take (csrc/gd4d_mha_dropout.h)."""
    from graph_detr4d_amd import fused_train, ops
    dec = G.build_transformer_layer_sequence(_decoder_cfg(dict(type='Deform3DCrossAttn', num_cams=6, pc_range=PC, num_points=4,
                                                               embed_dims=256, dropout=0.1)))
    layer = dec.layers[0]
    prm = fused_train._layer_params(layer)
    assert len(prm) == fused_train.PER_LAYER == len(fused_train.NAMES) == 32
    named = dict(layer.named_parameters())
    assert prm[0] is named['attentions.0.attn.in_proj_weight'] and prm[fused_train.NAMES.index('f1_b')] is named['ffns.0.layers.1.bias']
    assert {id(p) for p in prm} == {id(p) for p in named.values()}           # every parameter of the layer, once
    layer.eval()
    assert fused_train._dropouts(layer) == (0., 0., 0., 0., 0.)
    layer.train()
    assert fused_train._dropouts(layer) == pytest.approx((0.1, 0.1, 0.1, 0.1, 0.1))
    layer.attentions[0].proj_drop.p = 0.2                                     # proj_drop AND dropout_layer: two masks in a row
    assert fused_train._dropouts(layer) is None
    layer.attentions[0].proj_drop.p = 0.
    q = torch.zeros(5, 1, 256)
    assert not fused_train.applicable(dec, q, q, [torch.zeros(1, 6, 256, 4, 4)] * 4, torch.zeros(1, 5, 3), None, None,
                                      {id(layer.attentions[1]): (None, None, None, (object(), object()))}, (), dict(img_metas=[]))
    thresh, scale = ops.chain_dropout_args(0.1)
    assert thresh == round(0.1 * 2 ** 32) and scale == pytest.approx(1 / 0.9)
    assert ops.chain_dropout_args(0.)[0] == 0
    # the fragment arithmetic of the grouped image builder (include/gd4d.h: gd4d_image_job)
    assert ops.ImageJob.seg.size == 24 and ctypes_sizeof(ops.ImageJob) == 64


def ctypes_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)


def test_kv_plane_layouts_match_the_header():
    """ops.KVPlanes: the fragment layouts include/gd4d.h documents for GD4D_CHAIN_SPLIT_KV (K: [head][tile][lane = 16 (c / 8) + key %
    16][c % 8]; V: [head][step][d / 16][lane = 16 g + d % 16][j], key = 32 step + 16 (j >> 2) + 4 g + (j & 3)) and the row views the GPU
    tests compare against - built element by element from the formulas, on the CPU."""
    import torch
    from graph_detr4d_amd import ops
    m, c, heads = 37, 256, 8
    kv = ops.KVPlanes(m, c, 'cpu', heads=heads)
    assert kv.k.shape == (2, heads, 3, 64, 8) and kv.v.shape == (2, heads, 2, 2, 64, 8)
    rows_k = torch.arange(kv.tiles * 16 * c, dtype=torch.float32).view(kv.tiles * 16, c) % 251
    rows_v = (torch.arange(kv.steps * 32 * c, dtype=torch.float32).view(kv.steps * 32, c) * 3) % 241
    k = torch.zeros(heads, kv.tiles, 64, 8)
    for key in range(kv.tiles * 16):
        for ch in range(c):
            h, cc = divmod(ch, 32)
            k[h, key // 16, (cc // 8) * 16 + key % 16, cc % 8] = rows_k[key, ch]
    v = torch.zeros(heads, kv.steps, 2, 64, 8)
    for key in range(kv.steps * 32):
        step, ko = divmod(key, 32)
        t, kk = divmod(ko, 16)
        g, r = divmod(kk, 4)
        for ch in range(c):
            h, d = divmod(ch, 32)
            v[h, step, d // 16, g * 16 + d % 16, 4 * t + r] = rows_v[key, ch]
    kv.k[0], kv.v[0] = k.to(torch.bfloat16), v.to(torch.bfloat16)
    assert torch.equal(kv.k_rows()[0].float(), rows_k) and torch.equal(kv.v_rows()[0].float(), rows_v)


def test_point_padding_keeps_the_softmax_and_never_becomes_visible():
    """functional.pad_points (any num_points <= 8 on kernels compiled for 1 / 2 / 4 / 8): the padded points carry NaN offsets - every
    comparison of the visibility test (deform3d_cross_attn.py:239, :249-252) is false for them - and -inf logits, so the softmax over
    levels x points gives the real points exactly the weights it gives them without the padding; the padding's gradients are dropped."""
    import pytest
    from graph_detr4d_amd import functional as Fn
    from graph_detr4d_amd._lib import Gd4dError
    torch.manual_seed(0)
    b, q, hh, nl = 2, 7, 8, 4
    for p, tgt in ((1, 1), (2, 2), (3, 4), (4, 4), (5, 8), (6, 8), (8, 8)):
        off = torch.randn(b, q, hh, p, 3, requires_grad=True)
        lg = torch.randn(b, q, hh, nl, p, requires_grad=True)
        po, pl = Fn.pad_points(off, lg, (1, 2, 4, 8), 'test')
        assert po.shape == (b, q, hh, tgt, 3) and pl.shape == (b, q, hh, nl, tgt)
        assert torch.equal(po[..., :p, :], off) and torch.isnan(po[..., p:, :]).all()
        assert torch.equal(pl[..., :p], lg) and torch.isinf(pl[..., p:]).all() and (pl[..., p:] < 0).all()
        coords = po.detach()[..., 0]
        assert not ((coords[..., p:] > 0) & (coords[..., p:] < 1)).any(), 'a NaN coordinate passes no visibility comparison'
        w_pad = pl.detach().flatten(-2).softmax(-1).view(b, q, hh, nl, tgt)
        w = lg.detach().flatten(-2).softmax(-1).view(b, q, hh, nl, p)
        # (exactly 0 for the padding; the real points' weights differ from the unpadded softmax only by the order torch sums in -
        #  the kernel's own sum adds exact zeros)
        assert (w_pad[..., p:] == 0).all() and torch.allclose(w_pad[..., :p], w, rtol=1e-6, atol=1e-8)
        (po[..., :p, :].sum() + pl[..., :p].sum()).backward()
        assert torch.equal(off.grad, torch.ones_like(off)) and torch.equal(lg.grad, torch.ones_like(lg))
    with pytest.raises(Gd4dError):
        Fn.pad_points(torch.zeros(1, 1, 4, 5, 3), torch.zeros(1, 1, 4, 4, 5), (4,), 'test')


def test_torch_op_routes_are_explicit(monkeypatch):
    """functional.torch_ops_route: the kernels' route unless GD4D_TORCH_OPS=1 asks for torch ops; a shape the kernels do not cover
    raises and names the switch instead of taking another route silently."""
    import pytest
    from graph_detr4d_amd import functional as Fn
    from graph_detr4d_amd._lib import Gd4dError
    monkeypatch.delenv('GD4D_TORCH_OPS', raising=False)
    assert Fn.torch_ops_route('a module', True) is False
    with pytest.raises(Gd4dError, match='GD4D_TORCH_OPS'):
        Fn.torch_ops_route('a module with embed_dims = 512', False)
    monkeypatch.setenv('GD4D_TORCH_OPS', '1')
    assert Fn.torch_ops_route('a module', True) is True and Fn.torch_ops_route('a module', False) is True
    # ... and PER MODULE: the choice of one module does not re-route the others
    monkeypatch.delenv('GD4D_TORCH_OPS', raising=False)

    class M:
        pass
    a, b = M(), M()
    a.torch_ops = True
    assert Fn.torch_ops_route('a', False, module=a) is True and Fn.torch_ops_route('b', True, module=b) is False
    with Fn.torch_ops_for(b):
        assert Fn.torch_ops_route('b', False, module=b) is True
    with pytest.raises(Gd4dError, match='torch_ops'):
        Fn.torch_ops_route('b', False, module=b)


def test_camera_runs_of_the_position_embedding():
    """FeaturePositionEmbedding._runs: sorted camera indices -> the contiguous [a, e) runs the kernels are launched on."""
    from graph_detr4d_amd import FeaturePositionEmbedding as F
    assert F._runs([]) == []
    assert F._runs([3]) == [(3, 4)]
    assert F._runs([0, 1, 2, 5, 6, 9]) == [(0, 3), (5, 7), (9, 10)]
    assert F._runs(list(range(6, 24))) == [(6, 24)]

