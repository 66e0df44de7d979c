"""gd4d_value_proj_bwd_input / _bwd_weight (split-bf16 x3 MFMA) against fp64 contractions and against what autograd
derives for the reference's flatten / transpose / cat / Linear (deform3d_cross_attn.py:264-280).  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LEVELS = [[(16, 28), (8, 14), (4, 7), (2, 4)],      # rows 16-byte aligned at some levels only, ragged tails
          [(5, 13)], [(1, 1), (1, 3)], [(8, 8), (8, 8), (3, 3)],
          [(9, 40), (3, 11)]]                         # 360 px: several full 64-pixel tiles per camera row


def _case(levels, r=5, seed=3):
    torch.manual_seed(seed)
    feats = [torch.randn(r, 256, h, w) for h, w in levels]
    w = torch.randn(256, 256) * 0.06
    w[3, 7] = 1.0                      # asymmetric landmarks: catch transposed operands / outputs
    w[200, 1] = -2.0
    s = sum(h * ww for h, ww in levels)
    gout = torch.randn(r, s, 256)
    return feats, w, gout


def _ref(feats, w, gout):
    r = gout.shape[0]
    flat = torch.cat([f.reshape(r, 256, -1) for f in feats], 2).double()            # (R, C, S)
    gin = torch.matmul(gout.double(), w.double()).transpose(1, 2)                  # (R, C, S)
    parts = gin.split([f.shape[-2] * f.shape[-1] for f in feats], dim=2)
    gw = torch.einsum('rso,rcs->oc', gout.double(), flat)
    return [p.reshape(f.shape) for p, f in zip(parts, feats)], gw, gout.double().sum((0, 1))


@pytest.mark.parametrize('levels', LEVELS)
def test_bwd_input_matches_fp64(levels):
    from graph_detr4d_amd import ops
    feats, w, gout = _case(levels)
    want, _, _ = _ref(feats, w, gout)
    got = ops.value_proj_bwd_input(gout.cuda(), w.cuda(), levels)
    for g, x in zip(got, want):
        assert g.shape == x.shape
        assert (g.cpu().double() - x).abs().max().item() < 5e-5
    # accumulate=True adds to what the tensors hold (sum over decoder layers)
    base = [torch.randn_like(f).cuda() for f in feats]
    acc = [b.clone() for b in base]
    ops.value_proj_bwd_input(gout.cuda(), w.cuda(), levels, grads=acc, accumulate=True)
    for a, b, x in zip(acc, base, want):
        assert (a.cpu().double() - (b.cpu().double() + x)).abs().max().item() < 5e-5


@pytest.mark.parametrize('levels', LEVELS)
def test_bwd_weight_matches_fp64(levels):
    from graph_detr4d_amd import ops
    feats, w, gout = _case(levels, seed=4)
    _, want_w, want_b = _ref(feats, w, gout)
    gw, gb = ops.value_proj_bwd_weight(gout.cuda(), [f.cuda() for f in feats])
    scale = want_w.abs().max().item()
    assert (gw.cpu().double() - want_w).abs().max().item() < 2e-5 * scale + 1e-4
    assert (gb.cpu().double() - want_b).abs().max().item() < 1e-3
    gw2, none = ops.value_proj_bwd_weight(gout.cuda(), [f.cuda() for f in feats], want_bias=False)
    assert none is None and torch.equal(gw, gw2)          # fixed summation order: run-to-run identical


def test_autograd_function_matches_torch_autograd():
    """ValueProjFunction (HIP forward + HIP backward) against autograd of the reference's op sequence."""
    from graph_detr4d_amd.autograd import ValueProjFunction
    levels = [(12, 20), (6, 10), (3, 5)]
    feats, w, gout = _case(levels, r=4, seed=5)
    b = torch.randn(256)
    fa = [f.cuda().requires_grad_() for f in feats]
    wa, ba = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    out = ValueProjFunction.apply(wa, ba, *fa)
    out.backward(gout.cuda())
    fb = [f.cuda().requires_grad_() for f in feats]
    wb, bb = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    flat = torch.cat([f.flatten(2).transpose(1, 2) for f in fb], 1)
    torch.nn.functional.linear(flat, wb, bb).backward(gout.cuda())
    for x, y in zip(fa, fb):
        torch.testing.assert_close(x.grad, y.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(wa.grad, wb.grad, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(ba.grad, bb.grad, rtol=1e-4, atol=1e-3)


def test_bwd_full_size_against_fp64():
    """BASELINE size (24 cameras, 116x200 .. 15x25)."""
    from graph_detr4d_amd import ops, synthetic
    torch.manual_seed(7)
    dev = 'cuda'
    levels = list(synthetic.R50_LEVELS)
    feats = [torch.randn(24, 256, h, w, device=dev) for h, w in levels]
    w = torch.randn(256, 256, device=dev) * 0.06
    s = sum(h * ww for h, ww in levels)
    gout = torch.randn(24, s, 256, device=dev) * 0.01
    gw, gb = ops.value_proj_bwd_weight(gout, feats)
    gin = ops.value_proj_bwd_input(gout, w, levels)
    flat = torch.cat([f.reshape(24, 256, -1) for f in feats], 2)
    want_w = torch.zeros(256, 256, device=dev, dtype=torch.float64)
    for r in range(24):                                       # per camera: keeps the fp64 temporaries small
        want_w += gout[r].double().t() @ flat[r].double().t()
    scale = want_w.abs().max().item()
    assert (gw.double() - want_w).abs().max().item() < 2e-5 * scale
    assert (gb.double() - gout.double().sum((0, 1))).abs().max().item() < 1e-3
    start = 0
    for g, (h, ww) in zip(gin, levels):
        for r in (0, 11, 23):
            want = (gout[r, start:start + h * ww].double() @ w.double()).t().reshape(256, h, ww)
            assert (g[r].double() - want).abs().max().item() < 1e-6
        start += h * ww


def test_multi_layer_function_sums_the_pyramid_gradient():
    """ValueProjMultiFunction: NL layers over one pyramid, one backward; a layer whose output is unused gets no gradient."""
    from graph_detr4d_amd.autograd import ValueProjMultiFunction
    levels = [(12, 20), (6, 10), (3, 5)]
    feats, _, _ = _case(levels, r=4, seed=8)
    nl = 3
    ws = [torch.randn(256, 256) * 0.06 for _ in range(nl)]
    bs = [torch.randn(256) for _ in range(nl)]
    s = sum(h * w for h, w in levels)
    gouts = [torch.randn(4, s, 256) for _ in range(nl)]
    fa = [f.cuda().requires_grad_() for f in feats]
    wa = [w.cuda().requires_grad_() for w in ws]
    ba = [b.cuda().requires_grad_() for b in bs]
    outs = ValueProjMultiFunction.apply(nl, *wa, *ba, *fa)
    loss = sum((o * g.cuda()).sum() for o, g in zip(outs[:2], gouts[:2]))         # layer 2 unused
    loss.backward()
    fb = [f.cuda().requires_grad_() for f in feats]
    wb = [w.cuda().requires_grad_() for w in ws]
    bb = [b.cuda().requires_grad_() for b in bs]
    flat = torch.cat([f.flatten(2).transpose(1, 2) for f in fb], 1)
    sum((torch.nn.functional.linear(flat, wb[i], bb[i]) * gouts[i].cuda()).sum() for i in range(2)).backward()
    for x, y in zip(fa, fb):
        torch.testing.assert_close(x.grad, y.grad, rtol=1e-4, atol=2e-4)
    for i in range(2):
        torch.testing.assert_close(wa[i].grad, wb[i].grad, rtol=1e-4, atol=2e-3)
        torch.testing.assert_close(ba[i].grad, bb[i].grad, rtol=1e-4, atol=1e-3)
    assert wa[2].grad is None and ba[2].grad is None
