"""The decoder's training step on row chains (graph_detr4d_amd/fused_train.py) against the generic per-module autograd path
(GD4D_TRAIN_CHAINS=0: the path tests/test_training_gpu.py and tests/test_timed_size_parity_gpu.py pin against autograd of the
oracle), and the chain operations it adds (LN_BWD, GD4D_CHAIN_MASK_P2, stores from LOAD / ADD / SMALL_LINEAR, the grouped image
builder) against torch."""
import pytest
import torch
import torch.nn as nn

import graph_detr4d_amd as G
from golden_io import Golden
from graph_detr4d_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _transformer(g, return_intermediate=True, layers=None):
    m = g.meta
    n = m['num_cams']
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=layers or m['num_layers'], return_intermediate=return_intermediate,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                         embed_dims=256, dropout=0.1)],
                         feedforward_channels=512, ffn_dropout=0.1,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    if layers is None:
        tr.load_state_dict(g.state(), strict=True)
    return tr.to(DEV).eval()


def _reg_branches(nl, seed=5):
    torch.manual_seed(seed)
    return nn.ModuleList(nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 10))
                         for _ in range(nl)).to(DEV)


def _run(tr, g, reg, chains, monkeypatch, fuse=False, probe_seed=3, **kw):
    from graph_detr4d_amd import dist as D, fused_train
    monkeypatch.setenv('GD4D_TRAIN_CHAINS', '1' if chains else '0')
    calls = []
    real = fused_train.run
    monkeypatch.setattr(fused_train, 'run', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    qe = g.t('query_embed').to(DEV).clone().requires_grad_()
    feats = [f.to(DEV).clone().requires_grad_() for f in g.feats()]
    params = [p for p in tr.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    red = None
    if fuse:
        red = D.FlatGradAllReducer(params)
        red.bind(fuse_weight_grads=True)
        red.zero_grad()
    states, init_ref, inter_refs = tr(feats, qe, reg_branches=reg, img_metas=g.img_metas(), **kw)
    gen = torch.Generator().manual_seed(probe_seed)
    probe = torch.randn(states.shape, generator=gen).to(DEV)
    probe_r = torch.randn(inter_refs.shape, generator=gen).to(DEV)
    # (inter_refs carries a gradient exactly when there is no refinement - and so no detach, detr3d_transformer.py:199-214:
    #  the box loss of every level then reaches the reference points' Linear through inverse_sigmoid(inter_references))
    ((states * probe).sum() + (init_ref ** 2).sum() + (inter_refs * probe_r).sum()).backward()
    assert len(calls) == (1 if chains else 0)
    if fuse:
        grads = {n_: Dg.clone() for n_, Dg in zip([n_ for n_, p in tr.named_parameters() if p.requires_grad],
                                                  [red.view_of(p) if hasattr(red, 'view_of') else p.grad for p in params])}
        red.unfuse()
    else:
        grads = {n_: (None if p.grad is None else p.grad.clone()) for n_, p in tr.named_parameters() if p.requires_grad}
    return dict(states=states.detach(), refs=inter_refs.detach(), qe=qe.grad.clone(), feats=[f.grad.clone() for f in feats], params=grads)


def _compare(a, b, tol=2e-3):
    torch.testing.assert_close(a['states'], b['states'], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(a['refs'], b['refs'], rtol=1e-4, atol=1e-5)
    rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp_min(1e-12)).item()   # noqa: E731
    assert rel(a['qe'], b['qe']) < tol
    for x, y in zip(a['feats'], b['feats']):
        assert rel(x, y) < tol
    worst = ('', 0.)
    for k, y in b['params'].items():
        x = a['params'][k]
        if y is None:
            assert x is None or float(x.abs().max()) == 0., k
            continue
        assert x is not None, k
        r = rel(x, y)
        if r > worst[1]:
            worst = (k, r)
        assert r < tol, (k, r)
    return worst


@pytest.mark.parametrize('with_reg,intermediate', [(False, True), (True, True), (True, False)])
def test_chain_training_step_equals_the_generic_path(with_reg, intermediate, monkeypatch):
    g = Golden('decoder_deform')
    tr = _transformer(g, return_intermediate=intermediate)
    reg = _reg_branches(g.meta['num_layers']) if with_reg else None
    a = _run(tr, g, reg, True, monkeypatch)
    b = _run(tr, g, reg, False, monkeypatch)
    worst = _compare(a, b)
    print('largest relative parameter-gradient difference:', worst)


def test_chain_training_step_six_layers_with_refinement(monkeypatch):
    """Six randomly initialised layers (the shipped depth) with reg branches: every layer's refined points feed the next plan."""
    g = Golden('decoder_deform')
    torch.manual_seed(1)
    tr = _transformer(g, layers=6)
    reg = _reg_branches(6)
    a = _run(tr, g, reg, True, monkeypatch)
    b = _run(tr, g, reg, False, monkeypatch)
    _compare(a, b, tol=5e-3)


def test_chain_training_refuses_a_second_backward(monkeypatch):
    monkeypatch.setenv('GD4D_TRAIN_CHAINS', '1')
    g = Golden('decoder_deform')
    tr = _transformer(g)
    feats = [f.to(DEV).requires_grad_() for f in g.feats()]
    states, _, _ = tr(feats, g.t('query_embed').to(DEV), reg_branches=None, img_metas=g.img_metas())
    loss = (states ** 2).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='ONE backward'):
        loss.backward()


class _HashDropout(nn.Dropout):
    """nn.Dropout with the chains' mask: element (row, column) of a (.., C) tensor kept iff the hash of (seed, row C + column)
    says so (ops.chain_dropout_keep_mask) - puts the GENERIC path on the masks the chain path draws."""

    def __init__(self, p, seed):
        super().__init__(p)
        self.seed = seed

    def forward(self, x):
        if not self.training or self.p == 0.:
            return x
        n = x.shape[-1]
        keep = ops.chain_dropout_keep_mask(self.seed, x.numel() // n, n, self.p).view(x.shape)
        return x * keep / (1.0 - self.p)


@pytest.mark.parametrize('with_reg', [False, True])
def test_chain_training_step_in_train_mode_equals_the_generic_path_on_the_same_masks(with_reg, monkeypatch):
    """Modules in train mode (dropout 0.1 on the attention probabilities, after out_proj, after output_proj, inside and after the
    FFN - the reference's config): the chain path draws its masks from per-site seeds; the generic path is put on the SAME
    masks (its nn.Dropout modules replaced by _HashDropout, the attention core handed the same seeds) and must then agree in
    the outputs and in every gradient."""
    from graph_detr4d_amd import fused_train
    g = Golden('decoder_deform')
    tr = _transformer(g).train()
    nl = g.meta['num_layers']
    reg = _reg_branches(nl) if with_reg else None
    seeds = torch.randint(-2 ** 62, 2 ** 62, (5 * nl,), generator=torch.Generator().manual_seed(9), dtype=torch.int64).to(DEV)
    monkeypatch.setattr(fused_train, 'draw_seeds', lambda n, dev: seeds.clone())
    a = _run(tr, g, reg, True, monkeypatch)
    # generic path on the same masks
    for lid, layer in enumerate(tr.decoder.layers):
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        site = lambda i: seeds[5 * lid + i:5 * lid + i + 1]          # noqa: E731
        sa.dropout_layer = _HashDropout(sa.dropout_layer.p, site(1))
        ca.dropout = _HashDropout(ca.dropout.p, site(2))
        ffn.layers[0][2] = _HashDropout(ffn.layers[0][2].p, site(3))
        ffn.layers[2] = _HashDropout(ffn.layers[2].p, site(4))
    tr.train()
    order = [seeds[5 * lid:5 * lid + 1] for lid in range(nl)]
    monkeypatch.setattr(ops, 'mha_dropout_seed', lambda dev: order.pop(0))
    b = _run(tr, g, reg, False, monkeypatch)
    assert not order
    _compare(a, b)
    # and the masks matter: eval mode gives other outputs
    c_ = _run(tr.eval(), g, reg, True, monkeypatch)
    assert (c_['states'] - a['states']).abs().max() > 1e-2


def test_chain_gemm_dropout_and_dropmask_use_the_documented_mask():
    torch.manual_seed(4)
    m, p = 70, 0.3
    x, w, b = torch.randn(m, 256, device=DEV), torch.randn(512, 256, device=DEV) * 0.05, torch.randn(512, device=DEV)
    res = torch.randn(m, 512, device=DEV)
    seed = ops.mha_dropout_seed(torch.device(DEV))
    out, out0 = torch.empty(m, 512, device=DEV), torch.empty(m, 512, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, x), ops.chain_gemm(0, w, b, relu=True, out=out, add=res, dropout=(seed, p))], m)
    ops.row_chain_fwd([ops.chain_load(0, x), ops.chain_gemm(0, w, b, relu=True, out=out0)], m)
    keep = ops.chain_dropout_keep_mask(seed, m, 512, p)
    assert 0.6 < keep.float().mean().item() < 0.8
    torch.testing.assert_close(out, out0 * keep / (1 - p) + res, rtol=1e-6, atol=1e-6)
    gr = torch.randn(m, 512, device=DEV)
    masked = torch.empty(m, 512, device=DEV)
    ops.row_chain_fwd([ops.chain_load(1, gr), ops.chain_dropmask(1, 2, 512, seed, p, out=masked)], m)
    torch.testing.assert_close(masked, gr * keep / (1 - p), rtol=1e-6, atol=1e-7)


# ---- the chain operations ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('m,n,relu', [(900, 256, False), (37, 256, True), (16, 512, False), (100, 64, True)])
def test_chain_layernorm_bwd_matches_autograd(m, n, relu):
    torch.manual_seed(m + n)
    norm = nn.LayerNorm(n).to(DEV)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = torch.randn(m, n, device=DEV, requires_grad=True)
    gy = torch.randn(m, n, device=DEV)
    y = norm(x)
    y = torch.relu(y) if relu else y
    y.backward(gy)
    dx = torch.empty(m, n, device=DEV)
    part = torch.full(((m + 15) // 16 * 2 * n,), float('nan'), device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, gy), ops.chain_load(1, x.detach()),
                       ops.chain_layernorm_bwd(0, 1, norm, dst=0, relu=relu, out=dx, part=part)], m)
    torch.testing.assert_close(dx, x.grad, rtol=1e-4, atol=1e-5)
    # the forward's input straight from global memory (no LOAD operation): the same bits
    dx_g, part_g = torch.empty(m, n, device=DEV), torch.empty_like(part)
    ops.row_chain_fwd([ops.chain_load(0, gy), ops.chain_layernorm_bwd(0, x.detach(), norm, dst=0, relu=relu, out=dx_g, part=part_g)], m)
    assert torch.equal(dx_g, dx) and torch.equal(part_g, part)
    dg, db = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ops.layernorm_bwd_reduce_group([(part, (m, n), dg, db)], accumulate=False)
    torch.testing.assert_close(dg, norm.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db, norm.bias.grad, rtol=1e-4, atol=1e-4)
    # the same through the stand-alone kernel's workspace: one format
    dx2, ws, mc = ops.layernorm_bwd(x.detach(), norm.weight.detach(), norm.bias.detach(), gy, norm.eps, relu=relu, defer=True)
    assert ws.numel() == part.numel() * 4 and mc == (m, n)
    torch.testing.assert_close(dx, dx2, rtol=1e-5, atol=1e-6)


def test_chain_gemm_masked_by_a_relu_output():
    torch.manual_seed(0)
    m = 100
    w = torch.randn(256, 512, device=DEV) * 0.05                    # forward: y = h W^T, h (m, 512); backward: g_h = (g_y W) o [h > 0]
    h = torch.relu(torch.randn(m, 512, device=DEV))
    gy = torch.randn(m, 256, device=DEV)
    out = torch.empty(m, 512, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, gy), ops.chain_gemm(0, w.t().contiguous(), None, out=out, mask=h)], m)
    want = (gy.double() @ w.double()) * (h > 0)
    torch.testing.assert_close(out.double(), want, rtol=1e-4, atol=1e-4)
    out2 = torch.empty(m, 512, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, gy), ops.chain_gemm(0, w.t().contiguous(), None, out=out2, mask=h, mask_scale=1.25)], m)
    torch.testing.assert_close(out2, out * 1.25, rtol=1e-6, atol=1e-7)


def test_load_add_small_linear_store_their_rows():
    torch.manual_seed(2)
    m = 50
    a, b, c_ = (torch.randn(m, 256, device=DEV) for _ in range(3))
    o1, o2 = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    r = torch.rand(m, 3, device=DEV)
    w, bias = torch.randn(256, 3, device=DEV), torch.randn(256, device=DEV)
    o3 = torch.empty(m, 256, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, a, b, out=o1), ops.chain_add(1, 0, 256, add=c_, out=o2),
                       ops.chain_load(2, r, inv_sigmoid=True), ops.chain_small_linear(2, w, bias, 3, out=o3)], m)
    torch.testing.assert_close(o1, a + b)
    torch.testing.assert_close(o2, a + b + c_)
    isig = ops.inverse_sigmoid_fwd(r)
    torch.testing.assert_close(o3, isig @ w.t() + bias, rtol=1e-5, atol=1e-5)


def test_image_set_equals_single_images_and_survives_weight_updates():
    """Stacked, transposed (zero-padded) and exact images from ONE launch are byte-identical to gd4d_chain_weight_image of the
    explicitly built matrices; refresh() after an in-place update gives the new weights' images at the same addresses."""
    torch.manual_seed(3)
    w1, w2, w3 = torch.randn(24, 256, device=DEV), torch.randn(96, 256, device=DEV), torch.randn(128, 256, device=DEV)
    big = torch.randn(768, 256, device=DEV)
    b1, b2, b3 = torch.randn(24, device=DEV), torch.randn(96, device=DEV), torch.randn(128, device=DEV)
    s = ops.ImageSet(torch.device(DEV, torch.cuda.current_device()))
    i_stack, i_stack_t = s.add([w1, w2, w3]), s.add([w1, w2, w3], transposed=True)
    i_qk_t, i_v_t = s.add([big[:512]], transposed=True), s.add([big[512:]], transposed=True)
    i_exact = s.add([w2], exact=True)
    bias = s.add_concat([b1, b2, b3])
    ptrs = [i.img.data_ptr() for i in (i_stack, i_stack_t, i_qk_t, i_v_t, i_exact)]

    def check():
        s.refresh()
        stack = torch.cat([w1, w2, w3], 0)
        pad_t = torch.zeros(256, 256, device=DEV)
        pad_t[:, :248] = stack.t()
        assert torch.equal(i_stack.img, ops.chain_weight_image(stack.contiguous()))
        assert torch.equal(i_stack_t.img, ops.chain_weight_image(pad_t))
        assert (i_stack_t.n, i_stack_t.k) == (256, 256) and (i_stack.n, i_stack.k) == (248, 256)
        assert torch.equal(i_qk_t.img, ops.chain_weight_image(big[:512].t().contiguous()))
        assert torch.equal(i_v_t.img, ops.chain_weight_image(big[512:].t().contiguous()))
        assert torch.equal(i_exact.img, ops.chain_weight_image(w2.clone(), exact=True))
        assert torch.equal(bias, torch.cat([b1, b2, b3]))
    check()
    with torch.no_grad():
        for t in (w1, w2, w3, big, b1, b2, b3):
            t.mul_(1.5).add_(0.1)
    ops.invalidate_chain_images()
    check()
    assert ptrs == [i.img.data_ptr() for i in (i_stack, i_stack_t, i_qk_t, i_v_t, i_exact)]


def test_model_can_be_copied_and_saved_after_a_chain_training_step(monkeypatch, tmp_path):
    """The images of a decoder's weights (raw device pointers in a job table) are kept beside the module, not in it: an EMA
    hook's deepcopy and torch.save(model) after a training step work, and the copy trains on its own images."""
    import copy
    monkeypatch.setenv('GD4D_TRAIN_CHAINS', '1')
    g = Golden('decoder_deform')
    tr = _transformer(g)
    a = _run(tr, g, None, True, monkeypatch)
    twin = copy.deepcopy(tr)
    torch.save(tr, tmp_path / 'model.pt')
    with torch.no_grad():
        for p in twin.parameters():
            p.mul_(1.01)
    b = _run(twin, g, None, True, monkeypatch)
    c = _run(tr, g, None, True, monkeypatch)
    assert (b['states'] - a['states']).abs().max() > 1e-4              # the copy read ITS weights
    torch.testing.assert_close(c['states'], a['states'], rtol=0, atol=0)  # ... and the original still its own


@pytest.mark.parametrize('form', ['bool', 'float', 'list'])
def test_chain_training_step_with_a_self_attention_mask(form, monkeypatch):
    """H-DETR's decoder_self_attn_mask (h_detr3d_transformer.py:129-167: query groups must not see each other) in training: a
    (Q, Q) bool or additive float mask, or mmcv's [self-attention mask, cross-attention mask] list."""
    g = Golden('decoder_deform')
    tr = _transformer(g)
    q = g.meta['num_query']
    m = torch.zeros(q, q, dtype=torch.bool)
    m[: q // 2, q // 2:] = True
    m[q // 2:, : q // 2] = True
    m = m.to(DEV)
    mask = m if form == 'bool' else torch.zeros(q, q, device=DEV).masked_fill(m, -1e4) if form == 'float' else [m, None]
    a = _run(tr, g, None, True, monkeypatch, attn_masks=mask)
    b = _run(tr, g, None, False, monkeypatch, attn_masks=mask)
    _compare(a, b)
    c_ = _run(tr, g, None, True, monkeypatch)
    assert (c_['states'] - a['states']).abs().max() > 1e-3          # the mask matters


def test_backward_after_an_in_place_weight_update_is_refused(monkeypatch):
    """The backward chains read the images (W and W^T) the forward pass built: an optimizer step between the two is refused,
    as autograd refuses a saved tensor that was modified in place."""
    monkeypatch.setenv('GD4D_TRAIN_CHAINS', '1')
    g = Golden('decoder_deform')
    tr = _transformer(g)
    feats = [f.to(DEV).requires_grad_() for f in g.feats()]
    states, _, _ = tr(feats, g.t('query_embed').to(DEV), reg_branches=None, img_metas=g.img_metas())
    with torch.no_grad():
        tr.decoder.layers[0].ffns[0].layers[1].weight.mul_(1.5)
    with pytest.raises(RuntimeError, match='modified in place'):
        states.sum().backward()


def test_without_reg_branches_the_reference_points_keep_their_gradient(monkeypatch):
    """without reg branches the decoder returns the caller's reference points un-detached for every layer; the chain
    path must hand back a tensor autograd can follow (its Function marks its own copies non-differentiable), so that a loss on
    inter_references reaches the reference points' Linear exactly as on the per-module path."""
    g = Golden('decoder_deform')
    tr = _transformer(g)
    a = _run(tr, g, None, True, monkeypatch)
    b = _run(tr, g, None, False, monkeypatch)
    _compare(a, b)
    ga, gb = a['params']['reference_points.weight'], b['params']['reference_points.weight']
    assert gb.abs().max() > 0
    torch.testing.assert_close(ga, gb, rtol=2e-3, atol=2e-3 * float(gb.abs().max()))
    # ... and the probe on inter_references is what makes the difference visible: without it the two gradients differ from these
    g2 = Golden('decoder_deform')
    tr2 = _transformer(g2)
    qe = g2.t('query_embed').to(DEV).clone().requires_grad_()
    monkeypatch.setenv('GD4D_TRAIN_CHAINS', '1')
    states, init_ref, inter_refs = tr2([f.to(DEV).clone().requires_grad_() for f in g2.feats()], qe, reg_branches=None, img_metas=g2.img_metas())
    assert inter_refs.requires_grad


def test_the_step_without_hand_offs_is_the_same_step(monkeypatch):
    """GD4D_TRAIN_REG_BESIDE=0 (what a device that fails the XCD placement self-test gets): the reg branch closes chain B instead of
    running beside the next in-projection through a SIGNAL / WAIT hand-off - same outputs, same gradients, bit for bit."""
    from graph_detr4d_amd import ops
    g = Golden('decoder_deform')
    tr = _transformer(g)
    reg = _reg_branches(6)
    a = _run(tr, g, reg, True, monkeypatch)
    ops.check_handoff()
    monkeypatch.setenv('GD4D_TRAIN_REG_BESIDE', '0')
    b = _run(tr, g, reg, True, monkeypatch)
    assert torch.equal(a['states'], b['states']) and torch.equal(a['refs'], b['refs']) and torch.equal(a['qe'], b['qe'])
    for k in a['params']:
        if a['params'][k] is not None:
            assert torch.equal(a['params'][k], b['params'][k]), k
