"""Training backward on the RAW pyramid (csrc/gd4d_cross_attn_sliced_bwd.hip): gd4d_value_proj_heads_bwd,
gd4d_cross_attn_dot_sliced, gd4d_cross_attn_plan_bwd and the gd4d_pyramid_grad_* bucket sort against the projected-value
backward (gd4d_cross_attn_bwd + the value_proj chain rule in torch; itself checked against autograd of the oracle in
tests/test_training_gpu.py) and against fp64.  The module-level gradients against the oracle's autograd run through this
path by default (tests/test_training_gpu.py).  GPU only."""
import pytest
import torch

from graph_detr4d_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'
from graph_detr4d_amd import synthetic  # noqa: E402

PC_RANGE = synthetic.PC_RANGE


def _case(heads, levels, n, q, b, seed):
    gen = torch.Generator().manual_seed(seed)
    hw = [(24, 40), (12, 20), (6, 10), (3, 5)][:levels]
    feats = [torch.randn(b, n, 256, h, w, generator=gen) for h, w in hw]
    l2i = torch.from_numpy(synthetic.camera_rig((n + 5) // 6)[:n]).unsqueeze(0).repeat(b, 1, 1, 1)
    l2i = l2i + 0.01 * torch.randn(b, n, 4, 4, generator=gen) * (torch.arange(b).view(b, 1, 1, 1) > 0)
    ref = torch.rand(b, q, 3, generator=gen)
    offsets = torch.randn(b, q, heads, 4, 3, generator=gen) * 2.0
    attn = torch.randn(b, q, heads, levels, 4, generator=gen)
    cam = torch.randn(b, q, n, generator=gen)
    w_v = torch.randn(256, 256, generator=gen) / 16
    b_v = torch.randn(256, generator=gen)
    gout = torch.randn(b, q, 256, generator=gen)
    to = lambda t: t.to(DEV)
    return dict(hw=hw, feats=[to(f) for f in feats], l2i=to(l2i.contiguous()), ref=to(ref), offsets=to(offsets), attn=to(attn), cam=to(cam),
                w_v=to(w_v), b_v=to(b_v), gout=to(gout), heads=heads, b=b, n=n, q=q)


def _projected_backward(c, raw_cam=False):
    """The projected-value route: value = value_proj(pyramid) (torch), gd4d_cross_attn_bwd, chain rule back to the pyramid."""
    b, n, hh = c['b'], c['n'], c['heads']
    flat = torch.cat([f.flatten(3).permute(0, 1, 3, 2) for f in c['feats']], 2).reshape(b * n, -1, 256)     # (R, S, 256)
    val = (flat @ c['w_v'].t() + c['b_v']).view(b * n, -1, hh, 256 // hh).contiguous()
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    gv, gr, go, ga, gc = ops.cross_attn_bwd(val, c['hw'], c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], PC_RANGE, 900, 1600,
                                            c['gout'], query_order=order, raw_cam_weights=raw_cam)
    gflat = gv.view(b * n, -1, 256).double() @ c['w_v'].double()                                               # (R, S, 256)
    gfeats, s = [], 0
    for h, w in c['hw']:
        gfeats.append(gflat[:, s:s + h * w].permute(0, 2, 1).reshape(b * n, 256, h, w).float())
        s += h * w
    return gfeats, gr, go, ga, gc


def _raw_backward(c, layers=1, raw_cam=False):
    b, n, q, hh = c['b'], c['n'], c['q'], c['heads']
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    pyr = ops.PyramidView.slice_planar(sp, hw)
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    sink = ops.PyramidGrad(pyr, layers, b, q, hh)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    out = None
    for layer in range(layers):
        plan = ops.cross_attn_plan_fwd(pyr, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], PC_RANGE, 900, 1600, hh,
                                       query_order=order, raw_cam_weights=raw_cam)
        gagg, beta = ops.value_proj_heads_bwd(c['gout'], c['w_v'], c['b_v'], hh, grad_agg=sink.grad_agg_rows(layer))
        dpart = ops.cross_attn_dot_sliced(plan, gagg)
        out = ops.cross_attn_plan_bwd(plan, dpart, beta, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], PC_RANGE, 900, 1600,
                                      raw_cam_weights=raw_cam, status=status)
        sink.add_layer(layer, plan)
    gfeats = sink.finish()
    assert int(status.item()) == 0, 'an item count of the backward disagrees with the plan'
    return (gfeats,) + tuple(out)


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-12)).item()


def test_value_proj_heads_bwd_matches_fp64():
    gen = torch.Generator().manual_seed(3)
    for hh in (4, 8, 16):
        g = torch.randn(2, 37, 256, generator=gen).to(DEV)
        w = (torch.randn(256, 256, generator=gen) / 16).to(DEV)
        bias = torch.randn(256, generator=gen).to(DEV)
        gagg, beta = ops.value_proj_heads_bwd(g, w, bias, hh)
        dh = 256 // hh
        want = torch.einsum('bqhd,hdc->bqhc', g.double().view(2, 37, hh, dh), w.double().view(hh, dh, 256))
        want_b = (g.double().view(2, 37, hh, dh) * bias.double().view(hh, dh)).sum(-1)
        assert _rel(gagg, want) < 1e-5
        assert _rel(beta, want_b) < 1e-5
        _, beta0 = ops.value_proj_heads_bwd(g, w, None, hh)
        assert float(beta0.abs().max()) == 0.0


def test_value_proj_heads_bwd_weight_matches_fp64():
    gen = torch.Generator().manual_seed(4)
    for hh in (4, 8, 16):
        g = torch.randn(3, 301, 256, generator=gen).to(DEV)
        agg = torch.randn(3, 301, hh, 256, generator=gen).to(DEV)
        wsum = torch.rand(3, 301, hh, generator=gen).to(DEV)
        gw, gb = ops.value_proj_heads_bwd_weight(g, agg, wsum)
        dh = 256 // hh
        gd = g.double().view(-1, hh, dh)
        want = torch.einsum('mhd,mhc->hdc', gd, agg.double().view(-1, hh, 256)).reshape(256, 256)
        want_b = (gd * wsum.double().view(-1, hh, 1)).sum(0).reshape(256)
        assert _rel(gw, want) < 1e-5
        assert _rel(gb, want_b) < 1e-5
        gw2, none = ops.value_proj_heads_bwd_weight(g, agg, want_bias=False)
        assert none is None and torch.equal(gw2, gw)


@pytest.mark.parametrize('heads,levels,n,q,b', [(8, 4, 6, 96, 1), (8, 4, 24, 64, 1), (8, 4, 6, 50, 2), (8, 3, 6, 40, 1),
                                                 (4, 4, 6, 40, 1), (16, 2, 7, 33, 3), (8, 1, 12, 30, 1)])
def test_raw_backward_matches_projected_backward(heads, levels, n, q, b):
    c = _case(heads, levels, n, q, b, seed=11 + heads + levels + n + b)
    gf_p, gr_p, go_p, ga_p, gc_p = _projected_backward(c)
    gf_r, gr_r, go_r, ga_r, gc_r = _raw_backward(c)
    assert _rel(gr_r, gr_p) < 2e-4
    assert _rel(go_r, go_p) < 2e-4
    assert _rel(ga_r, ga_p) < 2e-4
    assert _rel(gc_r, gc_p) < 2e-4
    for a, e in zip(gf_r, gf_p):
        assert a.shape == e.shape
        assert _rel(a, e) < 2e-4
        # every pixel the projected route leaves at exactly zero (never sampled) is exactly zero here too
        assert float(a[e == 0].abs().max() if (e == 0).any() else 0.0) < 1e-6 * float(e.abs().max())


@pytest.mark.parametrize('n,q', [(6, 96), (24, 900), (6, 37)])
def test_record_count_in_the_forward_gathers_launch(n, q):
    """gd4d_cross_attn_agg_items_count_fwd: the forward gather of a training step hands out the records' slots as well - the
    aggregates are bit-identical to the gather alone, the per-chunk counts equal gd4d_pyramid_grad_count's, and the pyramid's
    gradient built from those slots equals the one built from the separate launch (another slot order: fp32 rounding)."""
    c = _case(8, 4, n, q, 1, seed=3 + n + q)
    b, hh = c['b'], c['heads']
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    pyr = ops.PyramidView.slice_planar(sp, hw)
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    got = {}
    for merged in (False, True):
        sink = ops.PyramidGrad(pyr, 2, b, q, hh)
        for layer in range(2):
            plan = ops.cross_attn_plan_fwd(pyr, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], PC_RANGE, 900, 1600, hh,
                                           query_order=order, both=True)
            if merged:
                agg = ops.cross_attn_agg_sliced_fwd(plan, count=(sink, layer))
                assert agg is not None
            else:
                agg = ops.cross_attn_agg_sliced_fwd(plan)
                sink.add_layer(layer, plan)
            ops.value_proj_heads_bwd(c['gout'], c['w_v'], c['b_v'], hh, grad_agg=sink.grad_agg_rows(layer))
        counts = sink.count.clone()
        got[merged] = (agg, plan.wsum.clone(), counts, sink.finish())
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    assert torch.equal(got[True][2], got[False][2]) and int(got[True][2].sum()) > 0
    for a, e in zip(got[True][3], got[False][3]):
        assert _rel(a, e) < 1e-5


def test_weight_gradients_ride_in_the_gather_dots_launch():
    """gd4d_cross_attn_dot_sliced_wgrad: the tiles of queued weight gradients are guest workgroups of a gather-dot - D is
    bit-identical to the gather-dot alone (over the passes the plan uses), the weight / bias gradients are added to their targets
    and equal an fp64 evaluation (and the 16-wave stand-alone kernel within fp32 rounding)."""
    c = _case(8, 4, 6, 96, 1, seed=31)
    hh = c['heads']
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    pyr = ops.PyramidView.slice_planar(sp, hw)
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    plan = ops.cross_attn_plan_fwd(pyr, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], PC_RANGE, 900, 1600, hh, query_order=order)
    gagg, _ = ops.value_proj_heads_bwd(c['gout'], c['w_v'], c['b_v'], hh)
    nb = ops.cross_attn_dot_bytes(1, 6, c['q'], hh)
    want = ops.cross_attn_dot_sliced(plan, gagg, dpart=torch.zeros(nb, device=DEV, dtype=torch.uint8))
    gen = torch.Generator().manual_seed(5)
    shapes = [(900, 256, 768, True), (900, 256, 256, True), (900, 512, 256, False), (900, 3, 256, True), (37, 256, 10, True)]
    probs, refs = [], []
    for m, k, n, bias in shapes:
        x, gy = torch.randn(m, k, generator=gen).to(DEV), torch.randn(m, n, generator=gen).to(DEV)
        gw, gb = torch.randn(n, k, generator=gen).to(DEV), (torch.randn(n, generator=gen).to(DEV) if bias else None)
        refs.append((gw.double() + gy.double().t() @ x.double(), None if gb is None else gb.double() + gy.double().sum(0)))
        probs.append((x, gy, gw, gb))
    alone = [(gw.clone(), None if gb is None else gb.clone()) for _, _, gw, gb in probs]
    ops.linear_bwd_weight_group([(x, gy, a[0], a[1]) for (x, gy, _, _), a in zip(probs, alone)], accumulate=True)
    got = ops.cross_attn_dot_sliced(plan, gagg, dpart=torch.zeros(nb, device=DEV, dtype=torch.uint8), wgrads=probs)
    assert torch.equal(got, want)
    for (x, gy, gw, gb), (rw, rb), (aw, ab) in zip(probs, refs, alone):
        assert _rel(gw, rw) < 2e-6 and _rel(gw, aw) < 2e-6
        if gb is not None:
            assert _rel(gb, rb) < 2e-6 and _rel(gb, ab) < 2e-6


def test_record_fills_ride_in_the_attention_backwards_launch():
    """gd4d_mha_core_bwd_fill: the fills of two (then one) layers are guest workgroups of the dk / dv launch - dq, dk, dv are
    bit-identical to gd4d_mha_core_bwd, and the pyramid's gradient equals the one of the stand-alone fills (another slot order: fp32
    rounding)."""
    c = _case(8, 4, 6, 96, 1, seed=21)
    b, q, hh = c['b'], c['q'], c['heads']
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    pyr = ops.PyramidView.slice_planar(sp, hw)
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    gen = torch.Generator().manual_seed(2)
    qkv = torch.randn(200, 1, 768, generator=gen).to(DEV)
    qh, kh, vh = qkv.split(256, dim=-1)
    do = torch.randn(200, 1, 256, generator=gen).to(DEV)
    out, lse = ops.mha_core_fwd(qh, kh, vh, 8, want_lse=True)
    want = ops.mha_core_bwd(qh, kh, vh, out, do, lse, 8, packed_qk=True)
    res = {}
    for riding in (False, True):
        sink = ops.PyramidGrad(pyr, 3, b, q, hh)
        for layer in range(3):
            plan = ops.cross_attn_plan_fwd(pyr, c['ref'], c['offsets'] * (1 + 0.1 * layer), c['attn'], c['cam'], c['l2i'], PC_RANGE,
                                           900, 1600, hh, query_order=order)
            sink.add_layer(layer, plan)
            ops.value_proj_heads_bwd(c['gout'], c['w_v'], c['b_v'], hh, grad_agg=sink.grad_agg_rows(layer))
        if riding:
            sink.scan()
            for n in (2, 1):
                got = ops.mha_core_bwd(qh, kh, vh, out, do, lse, 8, packed_qk=True, fills=sink.take_fills(n))
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
            assert sink.plans == [] and len(sink._riding) == 3 and sink.take_fills(2) is None
        res[riding] = sink.finish()
    for a, e in zip(res[True], res[False]):
        assert _rel(a, e) < 1e-5


def test_record_fills_ride_in_a_backward_chains_launch():
    """gd4d_row_chain_fill_fwd: the fills of two layers, then of one, are guest workgroups of a TRAINING chain launch (one program;
    then two programs) - the chain's outputs are bit-identical to the plain launches, and the pyramid's gradient equals the one of
    the stand-alone fills (the records are the same records: the slots were handed out by the count)."""
    c = _case(8, 4, 6, 96, 1, seed=23)
    b, q, hh = c['b'], c['q'], c['heads']
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    pyr = ops.PyramidView.slice_planar(sp, hw)
    order = ops.query_order_fwd(c['ref'], PC_RANGE)
    gen = torch.Generator().manual_seed(4)
    m = 200
    x, gy = torch.randn(m, 256, generator=gen).to(DEV), torch.randn(m, 256, generator=gen).to(DEV)
    w1, w2 = (torch.randn(256, 256, generator=gen) / 16).to(DEV), (torch.randn(512, 256, generator=gen) / 16).to(DEV)
    ln = torch.nn.LayerNorm(256).to(DEV)

    def programs(o1, o2, o3, part):
        # a backward-style program (a stored LOAD + a LayerNorm backward: the training instantiation) and a second, plain one
        pa = [ops.chain_load(0, gy, out=o3), ops.chain_layernorm_bwd(0, x, ln, dst=1, out=o1, part=part), ops.chain_gemm(1, w1, None, out=o2)]
        pb = [ops.chain_load(0, x), ops.chain_gemm(0, w2, None, out=None, dst=1)]
        return pa, pb
    blocks = (m + 15) // 16
    outs = [[torch.empty(m, 256, device=DEV) for _ in range(3)] + [torch.empty(blocks * 2 * 256, device=DEV)] for _ in range(3)]
    pa, pb = programs(*outs[0])
    ops.row_chain_fwd(pa, m)
    res = {}
    for riding in (False, True):
        sink = ops.PyramidGrad(pyr, 3, b, q, hh)
        for layer in range(3):
            plan = ops.cross_attn_plan_fwd(pyr, c['ref'], c['offsets'] * (1 + 0.1 * layer), c['attn'], c['cam'], c['l2i'], PC_RANGE,
                                           900, 1600, hh, query_order=order)
            sink.add_layer(layer, plan)
            ops.value_proj_heads_bwd(c['gout'], c['w_v'], c['b_v'], hh, grad_agg=sink.grad_agg_rows(layer))
        if riding:
            sink.scan()
            pa1, _ = programs(*outs[1])
            ops.row_chain_fwd(pa1, m, fills=sink.take_fills(2))
            pa2, pb2 = programs(*outs[2])
            ops.row_chain2_fwd(pa2, pb2, m, fills=sink.take_fills(1))
            assert sink.plans == [] and len(sink._riding) == 3 and sink.take_fills(2) is None
            for k in (1, 2):
                for a, e in zip(outs[k][:3], outs[0][:3]):
                    assert torch.equal(a, e)
        res[riding] = sink.finish()
    for a, e in zip(res[True], res[False]):
        assert _rel(a, e) < 1e-5


def test_raw_backward_raw_camera_weights():
    """GD4D_CA_RAW_CAM_WEIGHTS (Deform3DCrossAttnMP's neighbour pass): no sigmoid on the camera logits."""
    c = _case(8, 4, 6, 48, 1, seed=5)
    p, r = _projected_backward(c, raw_cam=True), _raw_backward(c, raw_cam=True)
    for a, e in zip(r[1:], p[1:]):
        assert _rel(a, e) < 2e-4
    for a, e in zip(r[0], p[0]):
        assert _rel(a, e) < 2e-4


def test_pyramid_gradient_sums_over_layers():
    """Three layers with the same inputs and grad rows: the one-pass gradient is three times a single layer's."""
    c = _case(8, 4, 6, 64, 1, seed=9)
    one = _raw_backward(c, layers=1)[0]
    three = _raw_backward(c, layers=3)[0]
    for a, e in zip(three, one):
        torch.testing.assert_close(a, 3 * e, rtol=1e-5, atol=1e-6 * float(e.abs().max()))


def test_query_side_gradients_are_bit_reproducible():
    """Every partial sum of gd4d_cross_attn_plan_bwd has its own slot and a fixed order: two runs agree bit for bit (the
    pyramid's gradient follows the atomic slot hand-out of the bucket fill, like the atomicAdd scatter it replaces)."""
    c = _case(8, 4, 12, 80, 1, seed=21)
    a, b = _raw_backward(c), _raw_backward(c)
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)


def test_full_size_raw_backward_matches_projected_backward():
    """900 queries x 24 cameras, R50 pyramid: the shapes bench.py --mode train runs."""
    gen = torch.Generator().manual_seed(0)
    n, q = 24, 900
    hw = synthetic.R50_LEVELS
    c = dict(hw=hw, feats=[torch.randn(1, n, 256, h, w, generator=gen).to(DEV) for h, w in hw],
             l2i=torch.from_numpy(synthetic.camera_rig(4)).unsqueeze(0).to(DEV), ref=torch.rand(1, q, 3, generator=gen).to(DEV),
             offsets=(torch.randn(1, q, 8, 4, 3, generator=gen) * 1.5).to(DEV), attn=torch.randn(1, q, 8, 4, 4, generator=gen).to(DEV),
             cam=torch.randn(1, q, n, generator=gen).to(DEV), w_v=(torch.randn(256, 256, generator=gen) / 16).to(DEV),
             b_v=torch.randn(256, generator=gen).to(DEV), gout=torch.randn(1, q, 256, generator=gen).to(DEV), heads=8, b=1, n=n, q=q)
    p, r = _projected_backward(c), _raw_backward(c)
    for a, e in zip(r[1:], p[1:]):
        assert _rel(a, e) < 5e-4
    for a, e in zip(r[0], p[0]):
        assert _rel(a, e) < 5e-4


@pytest.mark.parametrize('seed', [101, 102, 103, 104])
def test_raw_backward_random_seeds(seed):
    c = _case(8, 4, 12, 72, 1, seed=seed)
    p, r = _projected_backward(c), _raw_backward(c)
    for a, e in zip(r[1:], p[1:]):
        assert _rel(a, e) < 2e-4
    for a, e in zip(r[0], p[0]):
        assert _rel(a, e) < 2e-4


def test_raw_backward_edge_cases():
    """One camera, one query; and a step in which NO sample is visible (every reference point behind the cameras' image
    planes is impossible with a ring of cameras, so the points are pushed far above the rig): all gradients exactly zero."""
    c = _case(8, 4, 1, 1, 1, seed=7)
    p, r = _projected_backward(c), _raw_backward(c)
    for a, e in zip(r[1:], p[1:]):
        assert torch.allclose(a, e, rtol=2e-4, atol=2e-4 * float(e.abs().max()) + 1e-12)
    for a, e in zip(r[0], p[0]):
        assert torch.allclose(a, e, rtol=2e-4, atol=2e-4 * float(e.abs().max()) + 1e-12)
    c = _case(8, 4, 6, 40, 1, seed=8)
    c['offsets'] = torch.zeros_like(c['offsets'])
    c['offsets'][..., 2] = 500.0                       # 500 m above every camera: outside every image
    sp, hw = ops.pyramid_slice_planar_fwd(c['feats'])
    plan, mask = ops.cross_attn_plan_fwd(ops.PyramidView.slice_planar(sp, hw), c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'],
                                         PC_RANGE, 900, 1600, 8, want_mask=True)
    assert int(mask.sum().item()) == 0
    r = _raw_backward(c)
    for t in r[1:]:
        assert float(t.abs().max()) == 0.0
    for t in r[0]:
        assert float(t.abs().max()) == 0.0
