"""gd4d_linear_fwd / gd4d_layernorm_fwd / gd4d_mha_core_fwd against fp64 references.  GPU only."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('m,k,n', [(900, 256, 768), (900, 256, 248), (37, 3, 256), (900, 512, 256),
                                   (65, 256, 10), (1, 4, 5), (130, 36, 70)])
def test_linear_matches_fp64(m, k, n):
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    x, w, b = torch.randn(m, k), torch.randn(n, k) * 0.1, torch.randn(n)
    w[min(3, n - 1), min(2, k - 1)] = 2.5            # asymmetric landmark
    got = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda()).cpu()
    ref = F.linear(x.double(), w.double(), b.double())
    assert (got.double() - ref).abs().max().item() < 2e-5 * max(1.0, math.sqrt(k) / 4)


@pytest.mark.parametrize('k', [256, 1024])
@pytest.mark.parametrize('wkn', [False, True])
@pytest.mark.parametrize('x2', [False, True])
@pytest.mark.parametrize('res', [False, True])
def test_linear_branch_free_forms_match_fp64(k, wkn, x2, res):
    """The instantiations of linear_kernel for K % 64 == 0 (a chunk's loads issued as one batch): plain / transposed weight /
    addend on every column / residual operands, 4 waves and the 8-wave form of K >= 768, with row and column tails."""
    from graph_detr4d_amd import ops
    torch.manual_seed(k + 2 * wkn + 4 * x2 + 8 * res)
    for m, n in ((900, 256), (37, 24), (130, 1024)):
        x = torch.randn(m, k)
        xa = torch.randn(m, k) if x2 else None
        w = torch.randn(n, k) * 0.05
        w[min(3, n - 1), 2] = 2.5                                    # asymmetric landmark
        b = None if wkn else torch.randn(n)
        r1, r2 = (torch.randn(m, n), torch.randn(m, n)) if res else (None, None)
        wd = (w.t().contiguous() if wkn else w).cuda()                # weight_kn: the (K, N) layout
        c = lambda t: None if t is None else t.cuda()                # noqa: E731
        got = ops.linear_fwd(x.cuda(), wd, c(b), x2=c(xa), r1=c(r1), r2=c(r2), weight_kn=wkn).cpu()
        ref = F.linear((x + xa if x2 else x).double(), w.double(), None if b is None else b.double())
        if res:
            ref = ref + r1.double() + r2.double()
        assert got.shape == (m, n)
        assert (got.double() - ref).abs().max().item() < 2e-5 * max(1.0, math.sqrt(k) / 4) * (3 if res else 1)


def test_linear_fused_epilogue_and_input_addend():
    """y = relu((x + x2 [cols < 512]) W^T + b) + r1 + r2  - the in-projection / FFN / output_proj forms."""
    from graph_detr4d_amd import ops
    torch.manual_seed(0)
    m, k, n = 900, 256, 768
    x, x2 = torch.randn(m, 1, k), torch.randn(m, 1, k)
    w, b = torch.randn(n, k) * 0.06, torch.randn(n)
    r1, r2 = torch.randn(m, 1, n), torch.randn(m, 1, n)
    got = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), x2=x2.cuda(), n_split=512, relu=True,
                         r1=r1.cuda(), r2=r2.cuda()).cpu()
    xa = torch.cat([F.linear((x + x2).double(), w[:512].double(), b[:512].double()),
                    F.linear(x.double(), w[512:].double(), b[512:].double())], -1)
    ref = xa.relu() + r1.double() + r2.double()
    assert got.shape == (m, 1, n)
    assert (got.double() - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize('relu,res', [(False, False), (True, False), (False, True)])
def test_layernorm_matches_aten(relu, res):
    from graph_detr4d_amd import ops
    torch.manual_seed(1)
    x, r = torch.randn(900, 1, 256) * 3 + 0.5, torch.randn(900, 1, 256)
    g, b = torch.randn(256), torch.randn(256)
    got = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), res=r.cuda() if res else None, relu=relu).cpu()
    ref = F.layer_norm((x + r if res else x).double(), (256,), g.double(), b.double())
    if relu:
        ref = ref.relu()
    assert (got.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('l,b,mask', [(900, 1, None), (50, 2, None), (48, 1, 'bool'), (2700, 1, 'bool'),
                                      (77, 3, 'float'), (5, 1, None)])
def test_mha_core_matches_fp64(l, b, mask):
    from graph_detr4d_amd import ops
    torch.manual_seed(l)
    h, d = 8, 32
    qkv = torch.randn(l, b, 3 * h * d)
    am = None
    if mask == 'bool':
        kk = l // 3
        am = torch.zeros(l, l, dtype=torch.bool)
        am[kk:, :kk] = True
        am[:kk, kk:] = True
    elif mask == 'float':
        am = torch.randn(l, l)
    dq = qkv.cuda()
    qv, kv, vv = dq.split(h * d, dim=-1)
    got = ops.mha_core_fwd(qv, kv, vv, h, None if am is None else am.cuda()).cpu()
    q, k, v = (t.reshape(l, b * h, d).transpose(0, 1).double() for t in qkv.split(h * d, dim=-1))
    sc = torch.bmm(q / math.sqrt(d), k.transpose(1, 2))
    if am is not None:
        sc = sc.masked_fill(am, float('-inf')) if am.dtype == torch.bool else sc + am.double()
    ref = torch.bmm(sc.softmax(-1), v).transpose(0, 1).reshape(l, b, h * d)
    # inference: split-bf16 x3 products on the bf16 MFMA (~2^-16 relative per product: fp32-class, the arithmetic of the GEMMs
    # around the kernel); with the log-sum-exp asked for (training) the fp32-MFMA kernel runs: fp32-exact products
    assert (got.double() - ref).abs().max().item() < 1e-4
    assert (got.double() - ref).abs().median().item() < 5e-6
    got32, lse = ops.mha_core_fwd(qv, kv, vv, h, None if am is None else am.cuda(), want_lse=True)
    assert (got32.cpu().double() - ref).abs().max().item() < 2e-5
    assert (lse.cpu().double() - sc.logsumexp(-1).view(b, h, l).permute(2, 0, 1)).abs().max().item() < 1e-4


def test_linear_inverse_sigmoid_input():
    """position_encoder input transform: Linear(inverse_sigmoid(ref)), K = 3 and 4."""
    from graph_detr4d_amd import ops
    torch.manual_seed(2)
    for k in (3, 4):
        x = torch.rand(900, k)
        x[0] = 0.0
        x[1] = 1.0
        x[2] = 1e-7
        w, b = torch.randn(256, k), torch.randn(256)
        got = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), inv_sigmoid_in=True).cpu()
        xc = x.clamp(0, 1)
        isg = torch.log(xc.clamp(min=1e-5, max=1) / (1 - xc).clamp(min=1e-5, max=1))
        ref = F.linear(isg.double(), w.double(), b.double())
        assert (got.double() - ref).abs().max().item() < 5e-5


def test_refine_reference_matches_reference_formula():
    from graph_detr4d_amd import ops
    torch.manual_seed(9)
    tmp, ref = torch.randn(2, 900, 10), torch.rand(2, 900, 3)
    ref[0, 0] = torch.tensor([0.0, 1.0, 1e-7])
    got = ops.refine_reference_fwd(tmp.cuda(), ref.cuda()).cpu()

    def isg(x, eps=1e-5):
        x = x.clamp(0, 1)
        return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
    want = torch.zeros_like(ref)
    want[..., :2] = tmp[..., :2] + isg(ref[..., :2])
    want[..., 2:3] = tmp[..., 4:5] + isg(ref[..., 2:3])
    torch.testing.assert_close(got, want.sigmoid(), rtol=1e-5, atol=1e-6)


def test_refine_reference_order_matches_separate_kernels():
    from graph_detr4d_amd import ops
    g = torch.Generator().manual_seed(9)
    pc_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    for b, q in ((1, 900), (2, 1350), (1, 37)):
        tmp = torch.randn(b, q, 10, generator=g).cuda()
        ref = torch.rand(b, q, 3, generator=g).cuda()
        ref[0, 0] = torch.tensor([0., 1., 0.5])
        new_ref, order = ops.refine_reference_order_fwd(tmp, ref, pc_range)
        assert torch.equal(new_ref, ops.refine_reference_fwd(tmp, ref))
        assert torch.equal(torch.sort(order.cpu().long()).values, torch.arange(b * q))
        # same keys as the standalone order kernel: equal as multisets per azimuth bin -> compare sorted azimuths
        alone = ops.query_order_fwd(new_ref, pc_range).cpu().long()
        x = new_ref[..., 0].cpu().reshape(-1) * 102.4 - 51.2
        y = new_ref[..., 1].cpu().reshape(-1) * 102.4 - 51.2
        az = torch.atan2(y, x)
        both = torch.stack([az[order.cpu().long()], az[alone]])
        assert (both[0] - both[1]).abs().max().item() <= 2 * 3.1416 / (4096 // b) + 1e-5
    with pytest.raises(Exception):
        ops.query_order_fwd(torch.rand(1, 5000, 3).cuda(), pc_range)


def test_linear_group_equals_separate_launches():
    """gd4d_linear_group_fwd: the three Linears of (query + query_pos) in one launch, bit-identical to three."""
    from graph_detr4d_amd import ops
    torch.manual_seed(4)
    for m in (900, 37):
        x, x2 = torch.randn(m, 1, 256).cuda(), torch.randn(m, 1, 256).cuda()
        ws = [torch.randn(n, 256).cuda() * 0.1 for n in (24, 96, 128, 10)]
        bs = [torch.randn(24).cuda(), None, torch.randn(128).cuda(), torch.randn(10).cuda()]
        for g in (1, 3, 4):
            outs = ops.linear_group_fwd(x, ws[:g], bs[:g], x2=x2)
            for o, w, b in zip(outs, ws, bs):
                assert o.shape == (m, 1, w.shape[0])
                assert torch.equal(o, ops.linear_fwd(x, w, b, x2=x2))
        outs = ops.linear_group_fwd(x, ws[:2], bs[:2])
        assert torch.equal(outs[1], ops.linear_fwd(x, ws[1], None))
    from graph_detr4d_amd._lib import Gd4dError
    with pytest.raises(Gd4dError):
        ops.linear_group_fwd(x, ws + ws[:1], bs + bs[:1])


def test_linear_sum_output_and_strided_rows():
    """gd4d_linear_fwd / _group_fwd also hand out the x + x2 they form (kept by a training step for the weight gradients),
    and gd4d_linear_fwd reads a column slice of a wider buffer in place (row stride > K)."""
    from graph_detr4d_amd import ops
    torch.manual_seed(14)
    for m in (900, 37):
        x, x2 = torch.randn(m, 1, 256).cuda(), torch.randn(m, 1, 256).cuda()
        w, b = torch.randn(768, 256).cuda() * 0.1, torch.randn(768).cuda()
        y, xs = ops.linear_fwd(x, w, b, x2=x2, n_split=512, want_xsum=True)
        assert torch.equal(xs, x + x2) and torch.equal(y, ops.linear_fwd(x, w, b, x2=x2, n_split=512))
        ws = [torch.randn(n, 256).cuda() * 0.1 for n in (24, 96, 128)]
        outs, xs = ops.linear_group_fwd(x, ws, [None] * 3, x2=x2, want_xsum=True)
        assert torch.equal(xs, x + x2)
        assert all(torch.equal(o, o2) for o, o2 in zip(outs, ops.linear_group_fwd(x, ws, [None] * 3, x2=x2)))
        wide = torch.randn(m, 768).cuda()
        wk = torch.randn(512, 256).cuda()
        got = ops.linear_fwd(wide[:, :512], wk, weight_kn=True)
        assert torch.equal(got, ops.linear_fwd(wide[:, :512].contiguous(), wk, weight_kn=True))
        got = ops.linear_fwd(wide[:, 512:], w[:40], b[:40])
        assert torch.equal(got, ops.linear_fwd(wide[:, 512:].contiguous(), w[:40], b[:40]))
    with pytest.raises(ValueError):
        ops.linear_fwd(x, w, b, want_xsum=True)


@pytest.mark.parametrize('fused_main', [False, True])
def test_linear_group_function_gradients_match_fp64(fused_main):
    """autograd.LinearGroupFunction (the three Linears of query + query_pos as one node) against fp64 autograd of three
    F.linear calls: outputs, the shared input gradient, weight and bias gradients - returned to autograd, or added by the
    kernels to given buffers (the flat gradient buffer of a training step)."""
    from graph_detr4d_amd import functional as Fn
    torch.manual_seed(15)
    m = 900
    x = torch.randn(1, m, 256, device='cuda', requires_grad=True)
    x2 = torch.randn(1, m, 256, device='cuda', requires_grad=True)
    lins = [torch.nn.Linear(256, n).cuda() for n in (24, 96, 128)]
    bufs = []
    if fused_main:
        for lin in lins:
            for prm in (lin.weight, lin.bias):
                buf = torch.full_like(prm, 0.5)
                prm._gd4d_main_grad = buf
                bufs.append(buf)
    outs = Fn.linear_group_autograd(x, x2, lins)
    gos = [torch.randn_like(o) for o in outs]
    # the third output is used twice, the second not at all in one of the two passes
    (outs[0] * gos[0]).sum().backward(retain_graph=True)
    gx_first = x.grad.clone()
    x.grad = x2.grad = None
    for lin in lins:
        lin.weight.grad = lin.bias.grad = None
    for buf in bufs:
        buf.fill_(0.5)
    sum((o * g).sum() for o, g in zip(outs, gos)).backward()
    xd, x2d = x.detach().double().cpu().requires_grad_(True), x2.detach().double().cpu().requires_grad_(True)
    wd = [lin.weight.detach().double().cpu().requires_grad_(True) for lin in lins]
    bd = [lin.bias.detach().double().cpu().requires_grad_(True) for lin in lins]
    ref = [torch.nn.functional.linear(xd + x2d, w, b) for w, b in zip(wd, bd)]
    for o, r in zip(outs, ref):
        assert (o.detach().cpu().double() - r).abs().max().item() < 1e-4
    sum((r * g.cpu().double()).sum() for r, g in zip(ref, gos)).backward()
    assert (x.grad.cpu().double() - xd.grad).abs().max().item() < 2e-4
    assert torch.equal(x.grad, x2.grad)
    want_first = torch.autograd.grad((torch.nn.functional.linear(xd + x2d, wd[0], bd[0]) * gos[0].cpu().double()).sum(), xd)[0]
    assert (gx_first.cpu().double() - want_first).abs().max().item() < 2e-4   # (only one of the three outputs was used)
    for i, lin in enumerate(lins):
        if fused_main:
            gw, gb = bufs[2 * i] - 0.5, bufs[2 * i + 1] - 0.5
            assert lin.weight.grad is None and lin.bias.grad is None
        else:
            gw, gb = lin.weight.grad, lin.bias.grad
        assert (gw.cpu().double() - wd[i].grad).abs().max().item() < 2e-3
        assert (gb.cpu().double() - bd[i].grad).abs().max().item() < 2e-3


@pytest.mark.parametrize('k,isig,relu', [(3, True, True), (4, True, True), (2, False, False), (1, False, True)])
def test_small_linear_layernorm_matches_fp64(k, isig, relu):
    from graph_detr4d_amd import ops
    from oracle import torch_oracle as O
    torch.manual_seed(10 + k)
    m, c = 901, 256
    x = torch.rand(m, k) if isig else torch.randn(m, k)
    if isig:
        x[0] = 0.
        x[1] = 1.
    w, b = torch.randn(c, k), torch.randn(c)
    g_, be = torch.randn(c), torch.randn(c)
    got = ops.small_linear_layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), g_.cuda(), be.cuda(), relu=relu,
                                         inv_sigmoid_in=isig).cpu()
    xin = O.inverse_sigmoid(x) if isig else x
    ref = F.layer_norm(F.linear(xin.double(), w.double(), b.double()), (c,), g_.double(), be.double())
    if relu:
        ref = ref.relu()
    assert (got.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('m,k,n,ln,relu,relu2,res', [
    (900, 256, 256, True, False, False, 2), (900, 512, 256, True, False, False, 1), (900, 256, 256, True, False, True, 0),
    (37, 256, 256, True, True, False, 0), (900, 256, 768, False, False, False, 0), (900, 256, 512, False, True, False, 0),
    (901, 64, 200, True, False, False, 1), (17, 128, 10, False, False, False, 2)])
def test_linear_ln_matches_fp64(m, k, n, ln, relu, relu2, res):
    """gd4d_linear_ln_fwd against fp64: LN(act(x W^T + b) + r1 + r2) and the plain-Linear form."""
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    x, w, b = torch.randn(m, 1, k), torch.randn(n, k) * 0.08, torch.randn(n)
    r1 = torch.randn(m, 1, n) if res >= 1 else None
    r2 = torch.randn(m, 1, n) if res >= 2 else None
    gam, bet = (torch.randn(n), torch.randn(n)) if ln else (None, None)
    c = lambda t: None if t is None else t.cuda()
    got = ops.linear_ln_fwd(c(x), c(w), c(b), c(gam), c(bet), 1e-5, relu=relu, r1=c(r1), r2=c(r2),
                            relu_after_ln=relu2).cpu()
    ref = F.linear(x.double(), w.double(), b.double())
    if relu:
        ref = ref.relu()
    for r in (r1, r2):
        if r is not None:
            ref = ref + r.double()
    if ln:
        ref = F.layer_norm(ref, (n,), gam.double(), bet.double(), 1e-5)
    if relu2:
        ref = ref.relu()
    assert got.shape == (m, 1, n)
    assert (got.double() - ref).abs().max().item() < 5e-5


def test_linear_ln_input_addend_and_errors():
    from graph_detr4d_amd import ops
    from graph_detr4d_amd._lib import Gd4dError
    torch.manual_seed(1)
    m, k, n = 900, 256, 768
    x, x2 = torch.randn(m, 1, k).cuda(), torch.randn(m, 1, k).cuda()
    w, b = (torch.randn(n, k) * 0.06).cuda(), torch.randn(n).cuda()
    got = ops.linear_ln_fwd(x, w, b, x2=x2, n_split=512)
    ref = ops.linear_fwd(x, w, b, x2=x2, n_split=512)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)
    with pytest.raises(Gd4dError):                                   # LayerNorm over more than 256 columns
        ops.linear_ln_fwd(x, w, b, torch.ones(n).cuda(), torch.zeros(n).cuda())
    with pytest.raises(Gd4dError):                                   # K not a multiple of 64
        ops.linear_ln_fwd(torch.randn(5, 48).cuda(), torch.randn(8, 48).cuda())


@pytest.mark.parametrize('m,k,n', [(900, 256, 256), (900, 256, 24), (900, 4, 256), (900, 512, 256), (900, 256, 512),
                                   (7, 3, 5), (1801, 96, 10)])
def test_linear_bwd_weight_matches_fp64(m, k, n):
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    x, gy = torch.randn(m, k), torch.randn(m, n)
    gw, gb = ops.linear_bwd_weight(x.cuda(), gy.cuda())
    want_w, want_b = gy.double().t() @ x.double(), gy.double().sum(0)
    assert (gw.cpu().double() - want_w).abs().max().item() < 2e-4
    assert (gb.cpu().double() - want_b).abs().max().item() < 2e-4
    gw2, none = ops.linear_bwd_weight(x.cuda(), gy.cuda(), want_bias=False)
    assert none is None and torch.equal(gw, gw2)


def test_linear_function_gradients():
    from graph_detr4d_amd import functional as Fn
    torch.manual_seed(11)
    seq = torch.nn.Sequential(torch.nn.Linear(4, 256), torch.nn.LayerNorm(256), torch.nn.ReLU(inplace=True),
                              torch.nn.Linear(256, 256), torch.nn.ReLU(inplace=True), torch.nn.Linear(256, 10)).cuda()
    x = torch.randn(1, 900, 4, device='cuda', requires_grad=True)
    Fn.sequential_autograd(seq, x).square().sum().backward()
    got = [p.grad.clone() for p in seq.parameters()] + [x.grad.clone()]
    for p in seq.parameters():
        p.grad = None
    x.grad = None
    seq(x).square().sum().backward()
    for a, b in zip(got, [p.grad for p in seq.parameters()] + [x.grad]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * b.abs().max().item())


@pytest.mark.parametrize('m,k,n', [(900, 256, 256), (900, 512, 256), (900, 256, 512), (900, 10, 256), (37, 256, 3), (5, 7, 9)])
def test_linear_weight_kn_is_the_input_gradient(m, k, n):
    """GD4D_LIN_WEIGHT_KN: y = x W with W given (K, N) - the dgrad of a Linear whose weight is W."""
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    gy, w = torch.randn(m, k), torch.randn(k, n) * 0.1
    got = ops.linear_fwd(gy.cuda(), w.cuda(), weight_kn=True).cpu()
    ref = gy.double() @ w.double()
    assert got.shape == (m, n)
    assert (got.double() - ref).abs().max().item() < 2e-5 * max(1.0, math.sqrt(k) / 4)


@pytest.mark.parametrize('m,c,relu', [(900, 256, False), (900, 256, True), (33, 1024, False), (5, 12, True), (1801, 64, False)])
def test_layernorm_bwd_matches_fp64(m, c, relu):
    from graph_detr4d_amd import ops
    torch.manual_seed(m + c)
    x = (torch.randn(m, c) * 2 + 0.3)
    g, b, dy = torch.randn(c), torch.randn(c) * 0.5, torch.randn(m, c)
    dx, dg, db = ops.layernorm_bwd(x.cuda(), g.cuda(), b.cuda(), dy.cuda(), relu=relu)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.layer_norm(xd, (c,), gd, bd)
    if relu:
        y = y.relu()
    y.backward(dy.double())
    assert (dx.cpu().double() - xd.grad).abs().max().item() < 5e-5
    assert (dg.cpu().double() - gd.grad).abs().max().item() < 1e-3 * max(1.0, math.sqrt(m) / 30)
    assert (db.cpu().double() - bd.grad).abs().max().item() < 1e-3 * max(1.0, math.sqrt(m) / 30)
    # fixed summation order: run-to-run identical
    dx2, dg2, db2 = ops.layernorm_bwd(x.cuda(), g.cuda(), b.cuda(), dy.cuda(), relu=relu)
    assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dx, dx2)


@pytest.mark.parametrize('l,b,mask', [(900, 1, None), (50, 2, None), (48, 1, 'bool'), (2700, 1, 'bool'),
                                      (77, 3, 'float'), (5, 1, None)])
def test_mha_core_bwd_matches_fp64(l, b, mask):
    from graph_detr4d_amd import ops
    torch.manual_seed(l + 1)
    h, d = 8, 32
    qkv = torch.randn(l, b, 3 * h * d)
    do = torch.randn(l, b, h * d)
    am = None
    if mask == 'bool':
        kk = l // 3
        am = torch.zeros(l, l, dtype=torch.bool)
        am[kk:, :kk] = True
        am[:kk, kk:] = True
    elif mask == 'float':
        am = torch.randn(l, l)
    dev = qkv.cuda()
    qv, kv, vv = dev.split(h * d, dim=-1)                       # strided thirds of the packed in-projection
    amd = None if am is None else am.cuda()
    out, lse = ops.mha_core_fwd(qv, kv, vv, h, amd, want_lse=True)
    dq, dk, dv = ops.mha_core_bwd(qv, kv, vv, out, do.cuda(), lse, h, amd)
    ref = qkv.double().requires_grad_(True)
    q, k, v = (t.reshape(l, b * h, d).transpose(0, 1) for t in ref.split(h * d, dim=-1))
    sc = torch.bmm(q / math.sqrt(d), k.transpose(1, 2))
    if am is not None:
        sc = sc.masked_fill(am, float('-inf')) if am.dtype == torch.bool else sc + am.double()
    o = torch.bmm(sc.softmax(-1), v).transpose(0, 1).reshape(l, b, h * d)
    assert (out.cpu().double() - o).abs().max().item() < 2e-5
    assert (lse.cpu().double() - sc.logsumexp(-1).view(b, h, l).permute(2, 0, 1)).abs().max().item() < 1e-4
    o.backward(do.double())
    gq, gk, gv = ref.grad.split(h * d, dim=-1)
    tol = 1e-4 * max(1.0, math.sqrt(l) / 10)
    assert (dq.cpu().double() - gq).abs().max().item() < tol
    assert (dk.cpu().double() - gk).abs().max().item() < tol
    assert (dv.cpu().double() - gv).abs().max().item() < tol


@pytest.mark.parametrize('l,b,mask,p', [(900, 1, None, 0.1), (50, 2, 'bool', 0.5), (77, 3, 'float', 0.1), (2700, 1, 'bool', 0.1),
                                        (5, 1, None, 0.9)])
def test_mha_core_dropout_matches_fp64_with_the_same_mask(l, b, mask, p):
    """Dropout of the probabilities (nn.MultiheadAttention in training: F.dropout on the softmax output).  The kernels draw
    the keep decision of element (b, h, q, key) from a seed; ops.mha_dropout_keep_mask restates the draw with torch integer
    ops, and an fp64 softmax / dropout-by-that-mask / bmm gives the outputs and all three gradients to compare."""
    from graph_detr4d_amd import ops
    torch.manual_seed(l + 3)
    h, d = 8, 32
    qkv = torch.randn(l, b, 3 * h * d)
    do = torch.randn(l, b, h * d)
    am = None
    if mask == 'bool':
        kk = l // 3
        am = torch.zeros(l, l, dtype=torch.bool)
        am[kk:, :kk] = True
        am[:kk, kk:] = True
    elif mask == 'float':
        am = torch.randn(l, l)
    dev = qkv.cuda()
    qv, kv, vv = dev.split(h * d, dim=-1)
    amd = None if am is None else am.cuda()
    seed = ops.mha_dropout_seed('cuda')
    seed2 = ops.mha_dropout_seed('cuda')
    assert seed.item() != seed2.item() and (seed.item() >> 32) != (seed2.item() >> 32)
    out, lse = ops.mha_core_fwd(qv, kv, vv, h, amd, want_lse=True, dropout_p=p, seed=seed)
    dq, dk, dv = ops.mha_core_bwd(qv, kv, vv, out, do.cuda(), lse, h, amd, dropout_p=p, seed=seed)
    keep = ops.mha_dropout_keep_mask(seed, b, h, l, l, p).cpu()
    frac = keep.double().mean().item()
    assert abs(frac - (1 - p)) < 4 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-6, frac           # (4 sigma)
    keep2 = ops.mha_dropout_keep_mask(seed2, b, h, l, l, p).cpu()
    agree = (keep == keep2).double().mean().item()                 # independent draws agree with chance p^2 + (1-p)^2
    assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01 + 5 / math.sqrt(keep.numel())
    ref = qkv.double().requires_grad_(True)
    q, k, v = (t.reshape(l, b * h, d).transpose(0, 1) for t in ref.split(h * d, dim=-1))
    sc = torch.bmm(q / math.sqrt(d), k.transpose(1, 2))
    if am is not None:
        sc = sc.masked_fill(am, float('-inf')) if am.dtype == torch.bool else sc + am.double()
    pm = sc.softmax(-1) * keep.view(b * h, l, l).double() / (1 - p)
    o = torch.bmm(pm, v).transpose(0, 1).reshape(l, b, h * d)
    scale = 1 / (1 - p)
    assert (out.cpu().double() - o).abs().max().item() < 2e-5 * scale
    assert (lse.cpu().double() - sc.logsumexp(-1).view(b, h, l).permute(2, 0, 1)).abs().max().item() < 1e-4
    o.backward(do.double())
    gq, gk, gv = ref.grad.split(h * d, dim=-1)
    tol = 1e-4 * max(1.0, math.sqrt(l) / 10) * scale
    assert (dq.cpu().double() - gq).abs().max().item() < tol
    assert (dk.cpu().double() - gk).abs().max().item() < tol
    assert (dv.cpu().double() - gv).abs().max().item() < tol
    # a rate of zero is the plain kernel, bit for bit
    out0 = ops.mha_core_fwd(qv, kv, vv, h, amd)
    out00 = ops.mha_core_fwd(qv, kv, vv, h, amd, dropout_p=0., seed=seed)
    assert torch.equal(out0, out00)


def test_mha_core_dropout_with_different_query_and_key_counts():
    """Lq != Lk (the general MultiheadAttention call): the element ids of the dropout draw use both extents; forward and the
    two backward kernels against fp64 with the mask rebuilt on the host."""
    from graph_detr4d_amd import ops
    torch.manual_seed(21)
    lq, lk, b, h, d, p = 50, 77, 2, 8, 32, 0.3
    q = torch.randn(lq, b, h * d)
    k, v = torch.randn(lk, b, h * d), torch.randn(lk, b, h * d)
    do = torch.randn(lq, b, h * d)
    am = torch.randn(lq, lk)
    seed = ops.mha_dropout_seed('cuda')
    out, lse = ops.mha_core_fwd(q.cuda(), k.cuda(), v.cuda(), h, am.cuda(), want_lse=True, dropout_p=p, seed=seed)
    dq, dk, dv = ops.mha_core_bwd(q.cuda(), k.cuda(), v.cuda(), out, do.cuda(), lse, h, am.cuda(), dropout_p=p, seed=seed)
    keep = ops.mha_dropout_keep_mask(seed, b, h, lq, lk, p).cpu()
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    heads = lambda t, n: t.reshape(n, b * h, d).transpose(0, 1)       # noqa: E731
    sc = torch.bmm(heads(qd, lq) / math.sqrt(d), heads(kd, lk).transpose(1, 2)) + am.double()
    pm = sc.softmax(-1) * keep.view(b * h, lq, lk).double() / (1 - p)
    o = torch.bmm(pm, heads(vd, lk)).transpose(0, 1).reshape(lq, b, h * d)
    assert (out.cpu().double() - o).abs().max().item() < 4e-5
    o.backward(do.double())
    for got, want in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        assert (got.cpu().double() - want).abs().max().item() < 2e-4


def test_multihead_attention_train_mode_runs_the_hip_core_with_dropout(monkeypatch):
    """In train mode (attn_drop = 0.1, the reference's setting) the module keeps the core on the HIP kernels: the seed
    counter follows torch.manual_seed (same seed, same output and gradients), consecutive calls draw different masks, and
    the mean over many draws approaches the output without dropout (F.dropout's scaling)."""
    from graph_detr4d_amd import ops
    from graph_detr4d_amd.transformer_layers import MultiheadAttention
    calls = []
    real = ops.mha_core_fwd
    monkeypatch.setattr(ops, 'mha_core_fwd', lambda *a, **k: (calls.append(k.get('dropout_p')), real(*a, **k))[1])
    torch.manual_seed(5)
    mod = MultiheadAttention(256, 8, attn_drop=0.1, proj_drop=0.0).cuda()
    mod.train()
    q = torch.randn(300, 2, 256, device='cuda', requires_grad=True)
    pos = torch.randn(300, 2, 256, device='cuda')

    def run():
        q.grad = None
        mod.zero_grad()
        out = mod(q, query_pos=pos)
        out.square().sum().backward()
        return out.detach().clone(), q.grad.clone(), mod.attn.in_proj_weight.grad.clone()

    torch.manual_seed(11)
    a = run()
    b2 = run()
    torch.manual_seed(11)
    c = run()
    assert calls and all(abs(x - 0.1) < 1e-9 for x in calls)
    assert all(torch.equal(x, y) for x, y in zip(a, c))
    assert not torch.equal(a[0], b2[0])
    mod.eval()
    with torch.no_grad():
        want = mod(q.detach(), query_pos=pos)
    mod.train()
    acc = torch.zeros_like(want)
    n = 200
    with torch.enable_grad():
        for _ in range(n):
            acc += mod(q, query_pos=pos).detach()
    err = (acc / n - want).abs().max().item()
    one = (a[0] - want).abs().max().item()
    assert err < 0.25 * one, (err, one)                           # averaging 200 draws shrinks the deviation ~ 14 x


def test_multihead_attention_module_gradients_on_hip_kernels():
    """MultiheadAttention with autograd on: in-proj / core / out-proj all on gd4d kernels, against nn.MultiheadAttention."""
    from graph_detr4d_amd.transformer_layers import MultiheadAttention
    torch.manual_seed(5)
    mod = MultiheadAttention(256, 8, attn_drop=0.1, proj_drop=0.0, dropout_layer=dict(type='Dropout', drop_prob=0.1)).cuda()
    mod.eval()                                                   # dropout off: deterministic comparison
    q = torch.randn(900, 1, 256, device='cuda', requires_grad=True)
    pos = torch.randn(900, 1, 256, device='cuda')
    out = mod(q, query_pos=pos)
    out.square().sum().backward()
    got = [p.grad.clone() for p in mod.parameters()] + [q.grad.clone()]
    for p in mod.parameters():
        p.grad = None
    q.grad = None
    aten = torch.nn.MultiheadAttention(256, 8).cuda()
    aten.load_state_dict(mod.attn.state_dict())
    ref = q + aten(q + pos, q + pos, q)[0]
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    ref.square().sum().backward()
    names = [n for n, _ in mod.attn.named_parameters()]
    want = dict(aten.named_parameters())
    assert [n for n, _ in mod.named_parameters()] == ['attn.' + n for n in names]
    for a, b_ in zip(got, [want[n].grad for n in names] + [q.grad]):
        torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-4 * b_.abs().max().item())


def test_linear_bwd_weight_group_matches_the_single_launches():
    """gd4d_linear_bwd_weight_group: sixteen problems of different shapes in one launch, overwriting and accumulating."""
    from graph_detr4d_amd import ops
    gen = torch.Generator().manual_seed(5)
    shapes = [(900, 256, 256), (900, 256, 512), (900, 512, 256), (900, 256, 24), (900, 256, 96), (900, 256, 128), (900, 256, 768),
              (37, 64, 10), (900, 3, 256), (1, 256, 256), (450, 128, 33), (900, 256, 3), (64, 64, 64), (900, 256, 256), (128, 512, 17),
              (900, 40, 8)]
    probs, want = [], []
    for i, (m, k, n) in enumerate(shapes):
        x = torch.randn(m, k, generator=gen).to('cuda')
        gy = torch.randn(m, n, generator=gen).to('cuda')
        gw0 = torch.randn(n, k, generator=gen).to('cuda')
        gb0 = torch.randn(n, generator=gen).to('cuda') if i % 3 else None
        w1, b1 = ops.linear_bwd_weight(x, gy, want_bias=gb0 is not None)
        want.append((gw0 + w1, None if gb0 is None else gb0 + b1, w1, b1))
        probs.append((x, gy, gw0.clone(), None if gb0 is None else gb0.clone()))
    ops.linear_bwd_weight_group(probs, accumulate=True)
    for (x, gy, gw, gb), (ew, eb, _, _) in zip(probs, want):
        torch.testing.assert_close(gw, ew, rtol=1e-6, atol=1e-5)
        if gb is not None:
            torch.testing.assert_close(gb, eb, rtol=1e-6, atol=1e-5)
    ops.linear_bwd_weight_group(probs[:5], accumulate=False)
    for (x, gy, gw, gb), (_, _, w1, b1) in zip(probs[:5], want[:5]):
        assert torch.equal(gw, w1) and (gb is None or torch.equal(gb, b1))


def test_layernorm_bwd_deferred_column_reduce_matches_the_direct_call():
    """gd4d_layernorm_bwd with GD4D_LN_DEFER_REDUCE + gd4d_layernorm_bwd_reduce_group (32 problems, one launch): dx identical,
    dgamma / dbeta equal to the direct call's, overwriting and accumulating."""
    from graph_detr4d_amd import ops
    gen = torch.Generator().manual_seed(6)
    probs, want = [], []
    for i in range(32):
        m, c = [(900, 256), (900, 512), (37, 64), (1, 256), (450, 1024), (900, 4)][i % 6]
        x = torch.randn(m, c, generator=gen).to('cuda')
        gamma, beta = torch.randn(c, generator=gen).to('cuda'), torch.randn(c, generator=gen).to('cuda')
        gy = torch.randn(m, c, generator=gen).to('cuda')
        relu = i % 2 == 1
        dx0, dg0, db0 = ops.layernorm_bwd(x, gamma, beta, gy, 1e-5, relu=relu)
        dx1, ws, mc = ops.layernorm_bwd(x, gamma, beta, gy, 1e-5, relu=relu, defer=True)
        assert torch.equal(dx1, dx0) and mc == (m, c)
        g0, b0 = torch.randn(c, generator=gen).to('cuda'), torch.randn(c, generator=gen).to('cuda')
        probs.append((ws, mc, g0.clone(), b0.clone()))
        want.append((dg0, db0, g0, b0))
    ops.layernorm_bwd_reduce_group(probs, accumulate=True)
    for (_, _, g, b), (dg0, db0, g0, b0) in zip(probs, want):
        torch.testing.assert_close(g, g0 + dg0, rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(b, b0 + db0, rtol=1e-6, atol=1e-5)
    ops.layernorm_bwd_reduce_group(probs[:7], accumulate=False)
    for (_, _, g, b), (dg0, db0, _, _) in zip(probs[:7], want[:7]):
        assert torch.equal(g, dg0) and torch.equal(b, db0)
