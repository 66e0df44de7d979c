"""Training path: gradients of the Deform3DCrossAttn module and of a decoder layer (HIP gather forward +
backward, HIP value_proj forward) against torch autograd through the CPU oracle.  GPU only."""
import pytest
import torch

import graph_detr4d_amd as G
from golden_io import Golden, sub

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-8)).item()


@pytest.mark.parametrize('name', ['deform_n6', 'deform_n12_depth', 'deform_n24_b2'])
def test_cross_attn_module_gradients(name):
    """deform_n24_b2: the batch-2 fixture (the reference's row-pairing of value rows and logits, :277, in both directions)."""
    from oracle import torch_oracle as O
    g = Golden(name)
    m = g.meta
    mod = G.build_attention(dict(type='Deform3DCrossAttn', num_cams=m['num_cams'], pc_range=m['pc_range'],
                                 num_points=4, embed_dims=256, depth_encode=m['depth_encode']),
                            dict(batch_first=False))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()                      # eval(): dropout off, but autograd ON
    torch.manual_seed(1)
    gout = torch.randn_like(g.t('out'))

    # oracle gradients (CPU autograd)
    p_cpu = {k: v.clone().requires_grad_(True) for k, v in g.state().items()}
    q_cpu, qp_cpu = g.t('query').clone().requires_grad_(True), g.t('query_pos').clone().requires_grad_(True)
    ref_cpu = g.t('reference_points').clone().requires_grad_(True)
    feats_cpu = [f.clone().requires_grad_(True) for f in g.feats()]
    out_ref = O.deform3d_cross_attn(p_cpu, q_cpu, feats_cpu, qp_cpu, ref_cpu, g.img_metas(), m['pc_range'],
                                    m['num_heads'], m['num_points'], depth_encode=m['depth_encode'])
    (out_ref * gout).sum().backward()

    q, qp = g.t('query').to(DEV).requires_grad_(True), g.t('query_pos').to(DEV).requires_grad_(True)
    ref = g.t('reference_points').to(DEV).requires_grad_(True)
    feats = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    out = mod(q, None, feats, None, query_pos=qp, reference_points=ref, img_metas=g.img_metas())
    torch.testing.assert_close(out.detach().cpu(), out_ref.detach(), rtol=2e-4, atol=2e-4)
    (out * gout.to(DEV)).sum().backward()

    assert _rel(q.grad.cpu(), q_cpu.grad) < 2e-3
    assert _rel(qp.grad.cpu(), qp_cpu.grad) < 2e-3
    assert _rel(ref.grad.cpu(), ref_cpu.grad) < 2e-3
    for a, b in zip(feats, feats_cpu):
        assert _rel(a.grad.cpu(), b.grad) < 2e-3
    for k, prm in mod.named_parameters():
        assert prm.grad is not None, k
        assert _rel(prm.grad.cpu(), p_cpu[k].grad) < 3e-3, k


@pytest.mark.parametrize('name', ['deform_n6', 'deform_n24_b2'])
def test_channels_last_levels_train_in_place(name):
    """Levels stored (B, N, H, W, C) with autograd on: gathered in place forward (no copy of the pyramid), and their gradient
    comes back from gd4d_pyramid_grad_reduce in the same layout - values equal to the NCHW run's."""
    from graph_detr4d_amd import ops
    g = Golden(name)
    m = g.meta
    mod = G.build_attention(dict(type='Deform3DCrossAttn', num_cams=m['num_cams'], pc_range=m['pc_range'], num_points=4,
                                 embed_dims=256, depth_encode=m['depth_encode']), dict(batch_first=False))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()
    gout = torch.randn(g.t('out').shape, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run(feats):
        for p_ in mod.parameters():
            p_.grad = None
        q, qp = g.t('query').to(DEV).requires_grad_(True), g.t('query_pos').to(DEV).requires_grad_(True)
        out = mod(q, None, feats, None, query_pos=qp, reference_points=g.t('reference_points').to(DEV), img_metas=g.img_metas())
        (out * gout).sum().backward()
        return out.detach(), q.grad, {k: v.grad.clone() for k, v in mod.named_parameters()}
    nchw = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    nhwc = [f.to(DEV).permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3).requires_grad_(True) for f in g.feats()]
    assert all(ops.PyramidView.is_channels_last_level(f) for f in nhwc)
    o1, q1, p1 = run(nchw)
    o2, q2, p2 = run(nhwc)
    assert torch.equal(o1, o2)
    assert _rel(q2, q1) < 1e-5
    for k in p1:
        assert _rel(p2[k], p1[k]) < 1e-5, k
    for a, b in zip(nhwc, nchw):
        assert a.grad.shape == b.grad.shape and ops.PyramidView.is_channels_last_level(a.grad)
        assert _rel(a.grad, b.grad) < 1e-5


def test_bf16_value_storage_trains_on_the_raw_pyramid(monkeypatch):
    """value_dtype='bf16' with autograd on: the copy the gathers read is stored bf16 (features rounded, products and sums fp32);
    output and gradients stay within bf16 rounding of the fp32 module's."""
    monkeypatch.setenv('GD4D_TRAIN_VALUES', 'raw')           # (what the test is about, whatever the caller's environment says)
    g = Golden('deform_n6')
    m = g.meta
    outs = {}
    for vd in ('fp32', 'bf16'):
        mod = G.build_attention(dict(type='Deform3DCrossAttn', num_cams=m['num_cams'], pc_range=m['pc_range'], num_points=4,
                                     embed_dims=256, depth_encode=m['depth_encode'], value_dtype=vd), dict(batch_first=False))
        mod.load_state_dict(g.state(), strict=True)
        mod = mod.to(DEV).eval()
        q = g.t('query').to(DEV).requires_grad_(True)
        gen = torch.Generator().manual_seed(2)          # (the fixture's maps are bf16-representable: perturb them)
        feats = [(f + 1e-3 * torch.randn(f.shape, generator=gen)).to(DEV).requires_grad_(True) for f in g.feats()]
        out = mod(q, None, feats, None, query_pos=g.t('query_pos').to(DEV), reference_points=g.t('reference_points').to(DEV),
                  img_metas=g.img_metas())
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
        (out * gout).sum().backward()
        outs[vd] = (out.detach(), q.grad, [f.grad for f in feats], mod.value_proj.weight.grad)
    a, b = outs['bf16'], outs['fp32']
    assert not torch.equal(a[0], b[0])                   # the bf16 copy really was used
    assert _rel(a[0], b[0]) < 1e-2 and _rel(a[1], b[1]) < 2e-2 and _rel(a[3], b[3]) < 2e-2
    for x, y in zip(a[2], b[2]):
        assert _rel(x, y) < 2e-2


def test_raw_pyramid_path_refuses_a_second_backward_through_one_graph():
    g = Golden('deform_n6')
    m = g.meta
    mod = G.build_attention(dict(type='Deform3DCrossAttn', num_cams=m['num_cams'], pc_range=m['pc_range'], num_points=4,
                                 embed_dims=256, depth_encode=m['depth_encode']), dict(batch_first=False))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()
    feats = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    out = mod(g.t('query').to(DEV), None, feats, None, query_pos=g.t('query_pos').to(DEV),
              reference_points=g.t('reference_points').to(DEV), img_metas=g.img_metas())
    out.sum().backward(retain_graph=True)
    assert all(f.grad is not None for f in feats)
    with pytest.raises(RuntimeError, match='second backward'):
        out.sum().backward()


def test_decoder_layer_trains_one_step():
    """A full post-norm decoder layer in train() mode: loss decreases under SGD on the HIP path."""
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']
    layer = G.build_transformer_layer(dict(
        type='DetrTransformerDecoderLayer',
        attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                   dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4, embed_dims=256,
                        dropout=0.0)],
        feedforward_channels=512, ffn_dropout=0.0,
        operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))
    layer.load_state_dict(sub(g.state(), 'decoder.layers.0.'), strict=True)
    layer = layer.to(DEV).train()
    torch.manual_seed(2)
    qe = g.t('query_embed').to(DEV)
    query_pos, query = qe[:, :256].unsqueeze(1), qe[:, 256:].unsqueeze(1)
    ref = torch.rand(1, m['num_query'], 3, device=DEV)
    feats = [f.to(DEV) for f in g.feats()]
    target = torch.randn(m['num_query'], 1, 256, device=DEV)
    opt = torch.optim.SGD(layer.parameters(), lr=1e-2)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        out = layer(query, key=None, value=feats, query_pos=query_pos, reference_points=ref, img_metas=g.img_metas())
        loss = ((out - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]
    assert layer.attentions[1].value_proj.weight.grad.abs().max() > 0


def test_decoder_training_step_under_hipgraph_equals_eager():
    """Forward + backward of the 2-layer decoder (one autograd node for both layers' value_proj, gradients accumulated
    into the flat buffer dist.FlatGradAllReducer binds) captured in a hipGraph: replaying it gives the eager gradients."""
    from graph_detr4d_amd import dist as D
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=m['num_layers'], return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                         embed_dims=256, dropout=0.0)],
                         feedforward_channels=512, ffn_dropout=0.0,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    tr.load_state_dict(g.state(), strict=True)
    tr = tr.to(DEV).eval()
    qe = g.t('query_embed').to(DEV)
    feats = [f.to(DEV).requires_grad_() for f in g.feats()]
    metas = g.img_metas()
    params = [p for p in tr.parameters() if p.requires_grad]
    red = D.FlatGradAllReducer(params)
    red.bind()

    def step():
        red.zero_grad()
        states, _, _ = tr(feats, qe, reg_branches=None, img_metas=metas)
        (states ** 2).mean().backward()

    step()
    torch.cuda.synchronize()
    eager = red.flat.clone()
    eager_feat = [f.grad.clone() for f in feats]
    assert float(eager.abs().sum()) > 0
    for f in feats:
        f.grad = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    for f in feats:
        f.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    for _ in range(2):                                # replay after replay: nothing accumulates across steps
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(red.flat, eager, rtol=1e-4, atol=1e-6 * float(eager.abs().max()) + 1e-9)
        for f, e in zip(feats, eager_feat):
            torch.testing.assert_close(f.grad, e, rtol=1e-4, atol=1e-6 * float(e.abs().max()) + 1e-12)


def test_weight_gradients_accumulated_by_the_kernels_equal_autograds():
    """dist.FlatGradAllReducer.bind(fuse_weight_grads=True): the weight-gradient kernels (Linear, LayerNorm, value_proj of the
    aggregates, the row slices of the packed in-projection) add into the parameters' views of the flat buffer themselves and
    autograd gets nothing for those parameters - the buffer must end up with what autograd accumulates, and a second
    backward without zero_grad must double it."""
    from graph_detr4d_amd import dist as D
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=m['num_layers'], return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                         embed_dims=256, dropout=0.0)],
                         feedforward_channels=512, ffn_dropout=0.0,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    tr.load_state_dict(g.state(), strict=True)
    tr = tr.to(DEV).eval()
    qe = g.t('query_embed').to(DEV)
    feats = [f.to(DEV).requires_grad_() for f in g.feats()]
    params = [p for p in tr.parameters() if p.requires_grad]
    red = D.FlatGradAllReducer(params)

    def backward():
        states, _, _ = tr(feats, qe, reg_branches=None, img_metas=g.img_metas())
        (states ** 2).mean().backward()
    red.bind()
    red.zero_grad()
    backward()
    want = red.flat.clone()
    want_feats = [f.grad.clone() for f in feats]
    for f in feats:
        f.grad = None
    red.bind(fuse_weight_grads=True)
    assert all(getattr(p, '_gd4d_main_grad', None) is p.grad for p in params)
    red.zero_grad()
    backward()
    scale = float(want.abs().max())
    torch.testing.assert_close(red.flat, want, rtol=1e-5, atol=1e-6 * scale)
    for f, e in zip(feats, want_feats):
        assert ((f.grad - e).abs().max() / e.abs().max()).item() < 1e-5
    backward()                                        # no zero_grad in between: gradients accumulate
    torch.testing.assert_close(red.flat, 2 * want, rtol=1e-5, atol=2e-6 * scale)
    with pytest.raises(RuntimeError):
        red.install_hooks()
    red.unfuse()
    assert not any(hasattr(p, '_gd4d_main_grad') for p in params)


def test_queued_weight_gradients_of_a_module_applied_twice_do_not_race():
    """A parameter used by several nodes of one backward pass (the head's cls / reg branches are ONE module for all six levels
    when with_box_refine=False, detr3d_head_pe.py:410-413; tied weights): its queued weight-gradient entries must not share a
    grouped launch (the kernels add with a plain read-modify-write).  Gradients equal autograd's, run to run identical."""
    from graph_detr4d_amd import dist as D, functional as Fn
    torch.manual_seed(5)
    nn = torch.nn
    branch = nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(), nn.Linear(256, 10)).to(DEV)
    xs = [torch.randn(300, 256, device=DEV) for _ in range(6)]
    params = list(branch.parameters())

    def loss_of(apply):
        return sum((apply(x) ** 2).mean() * (i + 1) for i, x in enumerate(xs))
    loss_of(branch).backward()
    want = [p.grad.clone() for p in params]
    red = D.FlatGradAllReducer(params)
    red.bind(fuse_weight_grads=True)
    runs = []
    for _ in range(3):
        red.zero_grad()
        loss_of(lambda x: Fn.sequential_autograd(branch, x)).backward()
        runs.append([p.grad.clone() for p in params])
    red.unfuse()
    for g, w in zip(runs[0], want):
        torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-5 * float(w.abs().max()) + 1e-9)
    for other in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], other))


def test_decoder_trains_with_kernel_accumulated_and_queued_gradients():
    """Six SGD steps of the 2-layer decoder with every shortcut of the training step on (raw pyramid, weight gradients added to
    the flat buffer by the kernels, queued and grouped; pyramid trained too): the loss goes down, and the parameters end where
    plain autograd accumulation takes them."""
    from graph_detr4d_amd import dist as D
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']

    def build():
        tr = G.build_transformer(dict(
            type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
            decoder=dict(type='Detr3DTransformerDecoder', num_layers=m['num_layers'], return_intermediate=True,
                         transformerlayers=dict(
                             type='DetrTransformerDecoderLayer',
                             attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                                        dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                             embed_dims=256, dropout=0.0)],
                             feedforward_channels=512, ffn_dropout=0.0,
                             operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
        tr.load_state_dict(g.state(), strict=True)
        return tr.to(DEV).train()
    qe = g.t('query_embed').to(DEV)
    target = torch.randn(m['num_layers'], m['num_query'], 1, 256, generator=torch.Generator().manual_seed(11)).to(DEV)

    def train(fuse):
        tr = build()
        feats = [f.to(DEV).clone().requires_grad_() for f in g.feats()]
        params = [p for p in tr.parameters() if p.requires_grad]
        red = D.FlatGradAllReducer(params)
        red.bind(fuse_weight_grads=fuse)
        opt = torch.optim.SGD(params + feats, lr=2e-2)
        losses = []
        for _ in range(6):
            red.zero_grad()
            for f in feats:
                f.grad = None
            states, _, _ = tr(feats, qe, reg_branches=None, img_metas=g.img_metas())
            loss = ((states - target) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        if fuse:
            red.unfuse()
        return losses, torch.cat([p.detach().reshape(-1) for p in params]), torch.cat([f.detach().reshape(-1) for f in feats])
    l1, p1, f1 = train(True)
    l0, p0, f0 = train(False)
    assert l1[-1] < l1[0] and all(torch.isfinite(torch.tensor(l1)))
    assert abs(l1[-1] - l0[-1]) < 1e-4 * abs(l0[-1]) + 1e-7
    assert ((p1 - p0).abs().max() / p0.abs().max()).item() < 1e-5
    assert ((f1 - f0).abs().max() / f0.abs().max()).item() < 1e-5


def test_value_proj_weight_gradient_from_aggregates_equals_the_pixel_contraction(monkeypatch):
    """Decoder training, three routes to the same gradients: the raw-pyramid path (the default: plan + sliced gather forward,
    gd4d_cross_attn_sliced_bwd.hip backward, no projected value tensor), the projected-value path with value_proj's weight /
    bias gradient from the per-head aggregates (autograd.CrossAttnFunction), and the projected-value path with
    gd4d_value_proj_bwd_weight's contraction over every pixel row (what it does when the aggregates' kernels do not apply:
    functional.LateValues.applicable patched to say so)."""
    from oracle import torch_oracle as O
    g = Golden('decoder_deform')
    m = g.meta
    n = m['num_cams']
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=m['num_layers'], return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=m['pc_range'], num_points=4,
                                         embed_dims=256, dropout=0.0)],
                         feedforward_channels=512, ffn_dropout=0.0,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    tr.load_state_dict(g.state(), strict=True)
    tr = tr.to(DEV).train()
    feats = [f.to(DEV).requires_grad_() for f in g.feats()]
    qe = g.t('query_embed').to(DEV)
    probe = torch.randn(m['num_layers'], m['num_query'], 1, 256, generator=torch.Generator().manual_seed(7)).to(DEV)

    def grads(values, mode):
        from graph_detr4d_amd import functional as Fn
        monkeypatch.setenv('GD4D_TRAIN_VALUES', values)
        real = Fn.LateValues.applicable
        if mode == 'gemm':
            monkeypatch.setattr(Fn.LateValues, 'applicable', staticmethod(lambda *a, **k: False))
        else:
            monkeypatch.setattr(Fn.LateValues, 'applicable', real)
        for p_ in tr.parameters():
            p_.grad = None
        for f in feats:
            f.grad = None
        states, _, _ = tr(feats, qe, reg_branches=None, img_metas=g.img_metas())
        (states * probe).sum().backward()
        return {k: v.grad.clone() for k, v in tr.named_parameters() if v.grad is not None}, [f.grad.clone() for f in feats]
    ga, fa = grads('projected', 'agg')
    gg, fg = grads('projected', 'gemm')
    gr, fr = grads('raw', 'agg')           # the default: no projected value tensor at all (gd4d_cross_attn_sliced_bwd.hip)
    assert set(ga) == set(gg) == set(gr)
    for k in ga:
        scale = gg[k].abs().max().clamp(min=1e-6)
        assert ((ga[k] - gg[k]).abs().max() / scale).item() < (2e-3 if 'value_proj' in k else 1e-5), k
        assert ((gr[k] - gg[k]).abs().max() / scale).item() < (2e-3 if 'value_proj' in k else 5e-4), k
    for a, b, c in zip(fa, fg, fr):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        assert ((c - b).abs().max() / b.abs().max()).item() < 5e-4
    vp = [k for k in ga if 'value_proj' in k]
    assert len(vp) == 2 * m['num_layers'] and all(ga[k].abs().max() > 0 for k in vp)


@pytest.mark.parametrize('name,route', [('detr3d_n6', 'hip'), ('detr3d_n12_b2', 'hip'), ('detr3d_n6', 'torch')])
def test_detr3d_cross_atten_trains(name, route, monkeypatch):
    """The DETR3D baseline module with autograd on: output = the inference path's, gradients (query, query_pos, reference
    points, feature maps, every parameter) = autograd of the oracle (detr3d_transformer.py:352-438).  route hip: the sampling
    core on gd4d_detr3d_fwd / gd4d_detr3d_bwd (the default); torch: differentiable torch ops on the GPU."""
    from oracle import torch_oracle as O
    if route == 'torch':
        monkeypatch.setenv('GD4D_TORCH_OPS', '1')
    g = Golden(name)
    m = g.meta
    mod = G.build_attention(dict(type='Detr3DCrossAtten', num_cams=m['num_cams'], pc_range=m['pc_range'], num_points=1,
                                 embed_dims=256), dict(batch_first=False))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()
    with torch.no_grad():
        want = mod(g.t('query').to(DEV), None, [f.to(DEV) for f in g.feats()], None, query_pos=g.t('query_pos').to(DEV),
                   reference_points=g.t('reference_points').to(DEV), img_metas=g.img_metas())
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    p_cpu = {k: v.clone().requires_grad_(True) for k, v in g.state().items()}
    q_cpu, qp_cpu = g.t('query').clone().requires_grad_(True), g.t('query_pos').clone().requires_grad_(True)
    ref_cpu = g.t('reference_points').clone().requires_grad_(True)
    feats_cpu = [f.clone().requires_grad_(True) for f in g.feats()]
    out_ref = O.detr3d_cross_atten(p_cpu, q_cpu, feats_cpu, qp_cpu, ref_cpu, g.img_metas(), m['pc_range'])
    out_ref = out_ref[0] if isinstance(out_ref, tuple) else out_ref
    (out_ref * gout).sum().backward()
    q, qp = g.t('query').to(DEV).requires_grad_(True), g.t('query_pos').to(DEV).requires_grad_(True)
    ref = g.t('reference_points').to(DEV).requires_grad_(True)
    feats = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    out = mod(q, None, feats, None, query_pos=qp, reference_points=ref, img_metas=g.img_metas())
    torch.testing.assert_close(out.detach(), want, rtol=2e-4, atol=2e-4)
    (out * gout.to(DEV)).sum().backward()
    assert _rel(q.grad.cpu(), q_cpu.grad) < 2e-3
    assert _rel(qp.grad.cpu(), qp_cpu.grad) < 2e-3
    assert _rel(ref.grad.cpu(), ref_cpu.grad) < 5e-3
    for a, b in zip(feats, feats_cpu):
        assert _rel(a.grad.cpu(), b.grad) < 2e-3
    for k, prm in mod.named_parameters():
        assert prm.grad is not None, k
        assert _rel(prm.grad.cpu(), p_cpu[k].grad) < 3e-3, k


@pytest.mark.parametrize('route', ['raw', 'projected'])
@pytest.mark.parametrize('name', ['deform_mp_n6', 'deform_mp_n12_b2'])
def test_deform3d_cross_attn_mp_trains(name, route, monkeypatch):
    """Deform3DCrossAttnMP with autograd on (config detr4d_res50_deform_pe_mp...: the multi-point variant): gradients of the
    query, the reference points (centres and neighbours), the feature maps and every parameter that takes part against
    autograd of the oracle (deform3d_cross_attn_multi_point.py:196-453).  route 'raw' (default): both passes are
    CrossAttnRawFunction nodes on one slice-planar copy; 'projected': GD4D_TRAIN_VALUES=projected."""
    from oracle import torch_oracle as O
    from graph_detr4d_amd import ops
    if route == 'projected':
        monkeypatch.setenv('GD4D_TRAIN_VALUES', 'projected')
    pixel_rows = []
    real = ops.value_proj_fwd
    monkeypatch.setattr(ops, 'value_proj_fwd', lambda *a, **k: (pixel_rows.append(1), real(*a, **k))[1])
    g = Golden(name)
    m = g.meta
    mod = G.build_attention(dict(type='Deform3DCrossAttnMP', num_cams=m['num_cams'], pc_range=m['pc_range'],
                                 num_points=4, embed_dims=256))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()
    gout = torch.randn(g.t('out').shape, generator=torch.Generator().manual_seed(3))
    p_cpu = {k: v.clone().requires_grad_(True) for k, v in g.state().items()}
    q_cpu = g.t('query').clone().requires_grad_(True)
    ref_cpu = g.t('reference_points').clone().requires_grad_(True)
    feats_cpu = [f.clone().requires_grad_(True) for f in g.feats()]
    out_ref = O.deform3d_cross_attn_mp(p_cpu, q_cpu, feats_cpu, ref_cpu, g.img_metas(), m['pc_range'])
    out_ref = out_ref[0] if isinstance(out_ref, tuple) else out_ref
    torch.testing.assert_close(out_ref.detach(), g.t('out'), rtol=2e-4, atol=2e-4)
    (out_ref * gout).sum().backward()
    q = g.t('query').to(DEV).requires_grad_(True)
    ref = g.t('reference_points').to(DEV).requires_grad_(True)
    feats = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    out = mod(q, None, feats, reference_points=ref, img_metas=g.img_metas())
    torch.testing.assert_close(out.detach().cpu(), out_ref.detach(), rtol=2e-4, atol=2e-4)
    (out * gout.to(DEV)).sum().backward()
    assert _rel(q.grad.cpu(), q_cpu.grad) < 2e-3
    assert _rel(ref.grad.cpu(), ref_cpu.grad) < 5e-3
    for a, b in zip(feats, feats_cpu):
        assert _rel(a.grad.cpu(), b.grad) < 2e-3
    for k, prm in mod.named_parameters():
        if p_cpu[k].grad is None:                               # deform_sampling_offsets_neighbor: unused by the reference
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            continue
        assert prm.grad is not None, k
        assert _rel(prm.grad.cpu(), p_cpu[k].grad) < 3e-3, k
    assert bool(pixel_rows) == (route == 'projected')


@pytest.mark.parametrize('route', ['hip', 'torch'])
@pytest.mark.parametrize('name', ['detr3d_v2_n6', 'detr3d_v2_n12'])
def test_detr3d_cross_atten_v2_trains(name, route, monkeypatch):
    """Detr3DCrossAttenV2 with autograd on: output = the inference kernel's, gradients = autograd of the oracle
    (detr3d_transformer.py:441-710).  route 'hip': gd4d_detr3d_v2_fwd / gd4d_detr3d_v2_bwd behind one autograd node (the
    default); 'torch': the same sampling as differentiable torch ops (GD4D_TORCH_OPS=1)."""
    from oracle import torch_oracle as O
    from graph_detr4d_amd import ops
    if route == 'torch':
        monkeypatch.setenv('GD4D_TORCH_OPS', '1')
    calls = []
    real = ops.detr3d_v2_bwd
    monkeypatch.setattr(ops, 'detr3d_v2_bwd', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    g = Golden(name)
    m = g.meta
    mod = G.build_attention(dict(type='Detr3DCrossAttenV2', num_cams=m['num_cams'], pc_range=m['pc_range'],
                                 num_points=m['num_points'], embed_dims=256))
    mod.load_state_dict(g.state(), strict=True)
    mod = mod.to(DEV).eval()
    args = lambda t: dict(query_pos=t('query_pos'), reference_points=t('reference_points'))   # noqa: E731
    with torch.no_grad():
        want = mod(g.t('query').to(DEV), None, [f.to(DEV) for f in g.feats()], None, img_metas=g.img_metas(),
                   **{k: v.to(DEV) for k, v in args(g.t).items()})
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
    p_cpu = {k: v.clone().requires_grad_(True) for k, v in g.state().items()}
    q_cpu, qp_cpu = g.t('query').clone().requires_grad_(True), g.t('query_pos').clone().requires_grad_(True)
    ref_cpu = g.t('reference_points').clone().requires_grad_(True)
    feats_cpu = [f.clone().requires_grad_(True) for f in g.feats()]
    out_ref = O.detr3d_cross_atten_v2(p_cpu, q_cpu, feats_cpu, qp_cpu, ref_cpu, g.img_metas(), m['pc_range'],
                                      num_points=m['num_points'])
    (out_ref * gout).sum().backward()
    q, qp = g.t('query').to(DEV).requires_grad_(True), g.t('query_pos').to(DEV).requires_grad_(True)
    ref = g.t('reference_points').to(DEV).requires_grad_(True)
    feats = [f.to(DEV).requires_grad_(True) for f in g.feats()]
    out = mod(q, None, feats, None, query_pos=qp, reference_points=ref, img_metas=g.img_metas())
    torch.testing.assert_close(out.detach(), want, rtol=2e-4, atol=2e-4)
    (out * gout.to(DEV)).sum().backward()
    assert _rel(q.grad.cpu(), q_cpu.grad) < 2e-3
    assert _rel(qp.grad.cpu(), qp_cpu.grad) < 2e-3
    assert _rel(ref.grad.cpu(), ref_cpu.grad) < 5e-3
    for a, b in zip(feats, feats_cpu):
        assert _rel(a.grad.cpu(), b.grad) < 2e-3
    for k, prm in mod.named_parameters():
        assert prm.grad is not None, k
        assert _rel(prm.grad.cpu(), p_cpu[k].grad) < 3e-3, k
    assert len(calls) == (1 if route == 'hip' else 0)


@pytest.mark.parametrize('accumulate', [False, True])
def test_grouped_weight_gradients_match_fp64(accumulate):
    """gd4d_linear_bwd_weight_group on one call with the shapes a training step queues (256 / 512-wide Linears, the narrow camera /
    offset / position_encoder / branch layers, the packed in-projection), 900 and 37 rows, against fp64; accumulate adds to what
    the targets hold; the same bits on a second call."""
    from graph_detr4d_amd import ops
    gen = torch.Generator().manual_seed(17)
    shapes = [(900, 256, 256), (900, 256, 512), (900, 512, 256), (900, 256, 24), (900, 256, 96), (900, 3, 256), (37, 256, 128),
              (900, 256, 10), (900, 256, 768)]                       # (rows, inputs K, outputs N)
    probs, want = [], []
    for m, k, n in shapes:
        x, gy = torch.randn(m, k, generator=gen).to(DEV), torch.randn(m, n, generator=gen).to(DEV)
        gw0, gb0 = torch.randn(n, k, generator=gen).to(DEV), torch.randn(n, generator=gen).to(DEV)
        probs.append((x, gy, gw0.clone(), gb0.clone()))
        base_w, base_b = (gw0.double(), gb0.double()) if accumulate else (0, 0)
        want.append((gy.double().t() @ x.double() + base_w, gy.double().sum(0) + base_b,
                     (gy.double().abs().t() @ x.double().abs()).max().item()))
    ops.linear_bwd_weight_group(probs, accumulate=accumulate)
    for (x, gy, gw, gb), (ww, wb, scale) in zip(probs, want):
        assert (gw.double() - ww).abs().max().item() < 1.5e-5 * scale + 1e-5, (tuple(x.shape), tuple(gy.shape))
        torch.testing.assert_close(gb.double(), wb, rtol=1e-5, atol=2e-4)
    again = [(x, gy, torch.zeros_like(gw), torch.zeros_like(gb)) for x, gy, gw, gb in probs]
    once = [(x, gy, torch.zeros_like(gw), torch.zeros_like(gb)) for x, gy, gw, gb in probs]
    ops.linear_bwd_weight_group(again, accumulate=False)
    ops.linear_bwd_weight_group(once, accumulate=False)
    for a, b in zip(again, once):
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_value_proj_heads_weight_gradients_of_several_layers_in_one_launch():
    """gd4d_value_proj_heads_bwd_weight_group (what a training step queues per layer and issues once) equals the per-layer
    entry point bit for bit, problems with different row counts, with and without a bias target, accumulate on and off."""
    from graph_detr4d_amd import ops
    gen = torch.Generator().manual_seed(23)
    probs = []
    for m, bias in ((900, True), (900, True), (907, False), (37, True)):
        g_ = torch.randn(1, m, 256, generator=gen).to(DEV)
        agg, wsum = torch.randn(1, m, 8, 256, generator=gen).to(DEV), torch.rand(1, m, 8, generator=gen).to(DEV)
        probs.append((g_, agg, wsum, bias))
    for accumulate in (False, True):
        base_w = [torch.randn(256, 256, generator=gen).to(DEV) for _ in probs]
        base_b = [torch.randn(256, generator=gen).to(DEV) for _ in probs]
        one = [(w.clone(), b.clone()) for w, b in zip(base_w, base_b)]
        for (g_, agg, wsum, bias), (w, b) in zip(probs, one):
            if accumulate:
                ops.value_proj_heads_bwd_weight(g_, agg, wsum, want_bias=bias, into=(w, b if bias else None))
            else:
                gw, gb = ops.value_proj_heads_bwd_weight(g_, agg, wsum, want_bias=bias)
                w.copy_(gw)
                if bias:
                    b.copy_(gb)
        grp = [(w.clone(), b.clone()) for w, b in zip(base_w, base_b)]
        ops.value_proj_heads_bwd_weight_group([(g_, agg, wsum, w, b if bias else None) for (g_, agg, wsum, bias), (w, b) in zip(probs, grp)],
                                              accumulate=accumulate)
        for (a_w, a_b), (b_w, b_b), (_, _, _, bias) in zip(one, grp, probs):
            assert torch.equal(a_w, b_w)
            assert torch.equal(a_b, b_b)          # (untouched where the problem has no bias target)
