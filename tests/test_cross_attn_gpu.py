"""Parity of the fused HIP kernel (through the C ABI) against golden vectors captured from the
reference and against the CPU oracle on seeded inputs.  GPU only."""
import pytest
import torch

from golden_io import Golden

pytestmark = pytest.mark.gpu

DEFORM_CASES = ['deform_n6', 'deform_n12_depth', 'deform_n24_b2', 'deform_edge']
# fp32 tolerance of the path: north_star asks 1e-3; the kernel only reorders fp32 sums
ATOL = RTOL = 1e-4


def _inputs(g, dev, value_dtype=torch.float32):
    from oracle import torch_oracle as O
    m = g.meta
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    sd = g.state()
    flat, shapes = O.flatten_pyramid(g.feats())
    val = torch.nn.functional.linear(flat, sd['value_proj.weight'], sd['value_proj.bias'])
    val = val.view(b * n, -1, 8, 32).to(value_dtype)
    l2i = torch.from_numpy(g.arrays['lidar2img']).unsqueeze(0).expand(b, -1, -1, -1).contiguous()
    args = dict(value=val, level_hw=shapes, ref=g.t('reference_points'),
                offsets=g.t('offsets').view(b, q, 8, 4, 3).contiguous(),
                attn_logits=g.t('attn_logits').view(b, q, 8, 4, 4).contiguous(),
                cam_logits=g.t('cam_logits'), lidar2img=l2i)
    dev_args = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in args.items()}
    return args, dev_args, m


@pytest.mark.parametrize('head_major', [False, True])
@pytest.mark.parametrize('name', DEFORM_CASES)
def test_fused_kernel_matches_reference_golden(name, head_major):
    from graph_detr4d_amd import ops
    g = Golden(name)
    args, d, m = _inputs(g, 'cuda')
    if head_major:                         # (B*N, S, Hh, Dh) -> (B*N, Hh, S, Dh) planes
        d['value'] = d['value'].permute(0, 2, 1, 3).contiguous()
    out, mask, uv = ops.cross_attn_fwd(**d, pc_range=m['pc_range'], img_h=m['img_shape'][0],
                                       img_w=m['img_shape'][1], want_mask=True, want_uv=True,
                                       head_major=head_major)
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    gmask = g.t('mask').view(b, n, q, 8, 4, 4)[..., 0, :]          # (B,N,Q,Hh,P): equal for all levels
    guv = g.t('uv').view(b, n, q, 8, 4, 4, 2)[..., 0, :, :]
    assert torch.equal(mask.cpu(), gmask), 'visibility mask must be bit-exact'
    assert torch.equal(uv.cpu(), guv), 'projected coordinates must be bit-exact'
    torch.testing.assert_close(out.cpu(), g.t('agg'), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('head_major', [False, True])
@pytest.mark.parametrize('name', ['deform_n6', 'deform_n24_b2', 'deform_edge'])
def test_fused_kernel_bf16_values(name, head_major):
    """bf16 storage of the value tensor, fp32 accumulate, in both value layouts (head-major is the layout the
    module uses for bf16 storage - configs[1]): compare with the oracle fed the SAME bf16-rounded values (so only
    summation order differs)."""
    from graph_detr4d_amd import ops
    from oracle import torch_oracle as O
    g = Golden(name)
    args, d, m = _inputs(g, 'cuda', torch.bfloat16)
    if head_major:
        d['value'] = d['value'].permute(0, 2, 1, 3).contiguous()
    out = ops.cross_attn_fwd(**d, pc_range=m['pc_range'], img_h=m['img_shape'][0], img_w=m['img_shape'][1],
                             head_major=head_major)
    ref, _, _ = O.sample_aggregate(args['value'].float(), args['level_hw'], args['ref'], args['offsets'],
                                   args['attn_logits'].flatten(-2), args['cam_logits'], args['lidar2img'],
                                   m['pc_range'], m['img_shape'][0], m['img_shape'][1])
    torch.testing.assert_close(out.cpu(), ref, rtol=RTOL, atol=ATOL)


def test_fused_kernel_vs_oracle_seeded_midsize():
    """Seeded synthetic rig, Q=300, N=12, real pyramid aspect (reduced), random offsets."""
    from graph_detr4d_amd import ops, synthetic
    from oracle import torch_oracle as O
    torch.manual_seed(5)
    b, q, n = 1, 300, 12
    levels = [(29, 50), (15, 25), (8, 13), (4, 7)]
    l2i = torch.from_numpy(synthetic.camera_rig(2)).unsqueeze(0)
    val = torch.randn(b * n, sum(h * w for h, w in levels), 8, 32)
    ref = torch.rand(b, q, 3)
    offsets = torch.randn(b, q, 8, 4, 3) * 2.0
    attn = torch.randn(b, q, 8, 4, 4)
    cam = torch.randn(b, q, n)
    o_ref, uv_ref, m_ref = O.sample_aggregate(val, levels, ref, offsets, attn.flatten(-2), cam, l2i,
                                              synthetic.PC_RANGE, 900, 1600)
    out, mask = ops.cross_attn_fwd(val.cuda(), levels, ref.cuda(), offsets.cuda(), attn.cuda(), cam.cuda(),
                                   l2i.cuda(), synthetic.PC_RANGE, 900, 1600, want_mask=True)
    # torch's batched matmul on THIS host decides m_ref; the C-level arithmetic is pinned by the
    # golden tests above, so here only require agreement away from the decision thresholds.
    mism = (mask.cpu() != m_ref.to(torch.uint8))                 # (B, N, Q, Hh, P)
    assert mism.float().mean().item() < 1e-4
    # a flipped visibility bit changes ONE query row; every other row is compared, always
    flipped = mism.any(dim=4).any(dim=3).any(dim=1)              # (B, Q)
    assert flipped.sum().item() <= 3, f'{int(flipped.sum())} query rows with a flipped mask bit'
    keep = ~flipped
    torch.testing.assert_close(out.cpu()[keep], o_ref[keep], rtol=RTOL, atol=ATOL)


def test_linearity_and_invisible_cameras_full_size():
    """Size-independent properties at the BASELINE size (900 queries, 24 cameras, 4 levels):
    (i) linear in `value`; (ii) appending cameras that see nothing leaves the output unchanged
    given the same per-camera logits (T-invariance); (iii) all-zero value -> zero output."""
    from graph_detr4d_amd import ops, synthetic
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(11)
    b, q, n = 1, 900, 24
    levels = synthetic.R50_LEVELS
    s = sum(h * w for h, w in levels)
    l2i = torch.from_numpy(synthetic.camera_rig(4)).unsqueeze(0).to(dev)
    v1 = torch.randn(n, s, 8, 32, generator=g).to(dev)
    v2 = torch.randn(n, s, 8, 32, generator=g).to(dev)
    ref = torch.rand(b, q, 3, generator=g).to(dev)
    offsets = (torch.randn(b, q, 8, 4, 3, generator=g) * 2).to(dev)
    attn = torch.randn(b, q, 8, 4, 4, generator=g).to(dev)
    cam = torch.randn(b, q, n, generator=g).to(dev)

    def run(v, l2i_=l2i, cam_=cam):
        return ops.cross_attn_fwd(v, levels, ref, offsets, attn, cam_, l2i_, synthetic.PC_RANGE, 900, 1600)
    o1, o2, o12 = run(v1), run(v2), run(v1 * 0.5 + v2 * 2.0)
    torch.testing.assert_close(o12, o1 * 0.5 + o2 * 2.0, rtol=1e-4, atol=1e-4)
    assert run(torch.zeros_like(v1)).abs().max().item() == 0.0
    assert o1.abs().max().item() > 0.1
    # (ii) cameras behind everything: z <= 0 for every point => masked
    blind = l2i.clone()
    blind[:, 12:] = 0.0
    blind[:, 12:, 2, 3] = -1.0
    o_blind = run(v1, blind)
    o_half = ops.cross_attn_fwd(v1[:12].contiguous(), levels, ref, offsets, attn,
                                # scrambled view: camera n of query q reads flat[n*Q+q]; keep those values
                                cam.reshape(b, -1)[:, :12 * q].reshape(b, q, 12).contiguous(),
                                l2i[:, :12].contiguous(), synthetic.PC_RANGE, 900, 1600)
    torch.testing.assert_close(o_blind, o_half, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('heads,levels,n,q', [
    (8, [(6, 9)], 1, 7),                                   # one camera, one level, odd query count
    (8, [(12, 20), (6, 10)], 3, 33),
    (8, [(12, 20), (6, 10), (3, 5)], 6, 40),
    (8, [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 6, 21),   # 5 levels: runtime-L kernel path
    (4, [(12, 20), (6, 10), (3, 5), (2, 3)], 6, 50),       # Dh = 64
    (16, [(12, 20), (6, 10), (3, 5), (2, 3)], 6, 50),      # Dh = 16
    (8, [(6, 10), (3, 5)], 64, 9),                         # the 64-camera maximum
])
@pytest.mark.parametrize('head_major', [False, True])
def test_fused_kernel_shape_coverage_vs_c_oracle(heads, levels, n, q, head_major):
    """Every compiled template path against the plain-C oracle (bit-exact mask / uv), fp32 and bf16 storage, both
    value layouts."""
    import numpy as np
    from graph_detr4d_amd import ops
    from oracle import c_oracle
    rng = np.random.default_rng(heads * 1000 + n * 10 + len(levels))
    dh = 256 // heads
    s = sum(h * w for h, w in levels)
    val = rng.standard_normal((n, s, heads, dh)).astype(np.float32)
    ref = rng.random((1, q, 3)).astype(np.float32)
    ref[..., 0] = 0.5 + 0.2 * ref[..., 0]                  # keep a good share of points in view
    ref[..., 1] = 0.45 + 0.1 * ref[..., 1]
    offsets = (rng.standard_normal((1, q, heads, 4, 3)) * 1.0).astype(np.float32)
    attn = rng.standard_normal((1, q, heads, len(levels), 4)).astype(np.float32)
    cam = rng.standard_normal((1, q, n)).astype(np.float32)
    # n cameras fanned around the forward direction
    from graph_detr4d_amd import synthetic
    base = synthetic.camera_rig(max(1, (n + 5) // 6))[:n]
    l2i = base[None].astype(np.float32)
    pc = synthetic.PC_RANGE
    o_ref, m_ref, uv_ref = c_oracle.cross_attn_fwd(val, levels, ref, offsets, attn, cam, l2i, pc, 900, 1600)
    t = lambda a: torch.from_numpy(a).cuda()                # noqa: E731
    lay = (lambda v: v.permute(0, 2, 1, 3).contiguous()) if head_major else (lambda v: v)   # noqa: E731
    out, mask, uv = ops.cross_attn_fwd(lay(t(val)), levels, t(ref), t(offsets), t(attn), t(cam), t(l2i), pc, 900, 1600,
                                       want_mask=True, want_uv=True, head_major=head_major)
    assert m_ref.sum() > 0
    assert np.array_equal(mask.cpu().numpy(), m_ref)
    assert np.array_equal(uv.cpu().numpy(), uv_ref)
    np.testing.assert_allclose(out.cpu().numpy(), o_ref, rtol=1e-4, atol=1e-4)
    # bf16 storage of the same values
    vb = t(val).bfloat16()
    o_b, _, _ = c_oracle.cross_attn_fwd(vb.float().cpu().numpy(), levels, ref, offsets, attn, cam, l2i, pc, 900, 1600)
    out_b = ops.cross_attn_fwd(lay(vb), levels, t(ref), t(offsets), t(attn), t(cam), t(l2i), pc, 900, 1600,
                               head_major=head_major)
    np.testing.assert_allclose(out_b.cpu().numpy(), o_b, rtol=1e-4, atol=1e-4)


def test_unsupported_shapes_fail_loudly():
    from graph_detr4d_amd import _lib, ops
    z = lambda *s: torch.zeros(*s, device='cuda')           # noqa: E731
    with pytest.raises(_lib.Gd4dError, match='not supported'):   # 5 points per head
        ops.cross_attn_fwd(z(6, 4, 8, 32), [(2, 2)], z(1, 3, 3), z(1, 3, 8, 5, 3), z(1, 3, 8, 1, 5), z(1, 3, 6),
                           z(1, 6, 4, 4), [0, 0, 0, 1, 1, 1], 8, 8)
    with pytest.raises(_lib.Gd4dError, match='not supported'):   # embed_dims 128
        ops.cross_attn_fwd(z(6, 4, 8, 16), [(2, 2)], z(1, 3, 3), z(1, 3, 8, 4, 3), z(1, 3, 8, 1, 4), z(1, 3, 6),
                           z(1, 6, 4, 4), [0, 0, 0, 1, 1, 1], 8, 8)


@pytest.mark.parametrize('name', ['deform_n6', 'deform_n24_b2', 'deform_edge'])
def test_query_order_is_a_permutation_and_does_not_change_the_result(name):
    """gd4d_query_order_fwd only reschedules the workgroups: outputs, mask and uv stay bit-identical, for its own
    order and for an arbitrary permutation."""
    from graph_detr4d_amd import ops
    g = Golden(name)
    args, d, m = _inputs(g, 'cuda')
    kw = dict(pc_range=m['pc_range'], img_h=m['img_shape'][0], img_w=m['img_shape'][1])
    base = ops.cross_attn_fwd(**d, **kw, want_mask=True, want_uv=True)
    bq = m['batch'] * m['num_query']
    order = ops.query_order_fwd(d['ref'], m['pc_range'])
    assert order.dtype == torch.int32 and order.shape == (bq,)
    assert torch.equal(torch.sort(order.cpu().long()).values, torch.arange(bq))
    perm = torch.randperm(bq, generator=torch.Generator().manual_seed(5)).int().cuda()
    for o in (order, perm):
        got = ops.cross_attn_fwd(**d, **kw, want_mask=True, want_uv=True, query_order=o)
        for a, b_ in zip(got, base):
            assert torch.equal(a, b_)
    with pytest.raises(ValueError):
        ops.cross_attn_fwd(**d, **kw, query_order=order[:-1])
    with pytest.raises(ValueError):
        ops.cross_attn_fwd(**d, **kw, query_order=order.long())


def test_query_order_sorts_by_sample_then_azimuth():
    """Keys are (sample, azimuth bin of the de-normalised reference point about the lidar origin)."""
    import math
    from graph_detr4d_amd import ops
    g = torch.Generator().manual_seed(3)
    b, q = 2, 700
    pc_range = [-40., -30., -5., 60., 50., 3.]                    # origin not at the centre of the range
    ref = torch.rand(b, q, 3, generator=g)
    order = ops.query_order_fwd(ref.cuda(), pc_range).cpu().long()
    assert torch.equal(torch.sort(order).values, torch.arange(b * q))
    sample = order // q
    assert bool((sample[1:] >= sample[:-1]).all())
    x = ref[..., 0] * 100. - 40.
    y = ref[..., 1] * 80. - 30.
    az = torch.atan2(y, x).reshape(-1)[order]
    for s_ in range(b):
        a = az[sample == s_]
        # non-decreasing up to the width of one bin (2 pi / 2048) and fp32 rounding
        assert bool((a[1:] - a[:-1] > -(2 * math.pi / 2048) - 1e-5).all())


@pytest.mark.parametrize('b,n,q,raw', [(1, 6, 37, False), (2, 12, 20, True), (1, 24, 64, True)])
def test_one_point_per_level_and_raw_camera_weights(b, n, q, raw):
    """P = 1 form of the fused kernel (no offsets needed, one sample per level) and GD4D_CA_RAW_CAM_WEIGHTS - the
    neighbour pass of Deform3DCrossAttnMP - against the oracle."""
    from graph_detr4d_amd import ops, synthetic
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(100 * b + n + q)
    levels = [(16, 28), (8, 14), (4, 7), (2, 4)]
    s = sum(h * w for h, w in levels)
    val = torch.randn(b * n, s, 8, 32, generator=g)
    rig = torch.from_numpy(synthetic.camera_rig(n // 6, (128, 224))).unsqueeze(0).expand(b, -1, -1, -1).contiguous()
    ref = torch.rand(b, q, 3, generator=g)
    off = torch.randn(b, q, 8, 1, 3, generator=g)
    attn = torch.randn(b, q, 8, 4, 1, generator=g)
    cam = torch.randn(b, q, n, generator=g)
    c = lambda t: t.cuda()
    out, mask = ops.cross_attn_fwd(c(val), levels, c(ref), c(off), c(attn), c(cam), c(rig), synthetic.PC_RANGE, 128, 224,
                                   want_mask=True, raw_cam_weights=raw)
    exp, _, emask = O.sample_aggregate(val, levels, ref, off, attn.view(b, q, 8, 4), cam, rig, synthetic.PC_RANGE,
                                       128, 224, raw_cam=raw)
    assert torch.equal(mask.cpu().bool(), emask)
    torch.testing.assert_close(out.cpu(), exp, rtol=RTOL, atol=ATOL)
