"""gd4d_cross_attn_bwd against torch autograd through the CPU oracle's sample_aggregate
(same maths as the reference's autograd path: grid_sample backward + elementwise chain).  GPU only."""
import pytest
import torch

from golden_io import Golden

pytestmark = pytest.mark.gpu


def _case(name):
    from oracle import torch_oracle as O
    g = Golden(name)
    m = g.meta
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    sd = g.state()
    flat, shapes = O.flatten_pyramid(g.feats())
    val = torch.nn.functional.linear(flat, sd['value_proj.weight'], sd['value_proj.bias']).view(b * n, -1, 8, 32)
    l2i = torch.from_numpy(g.arrays['lidar2img']).unsqueeze(0).expand(b, -1, -1, -1).contiguous()
    t = dict(value=val.contiguous(), ref=g.t('reference_points').clone(),
             offsets=g.t('offsets').view(b, q, 8, 4, 3).clone(), attn=g.t('attn_logits').view(b, q, 8, 16).clone(),
             cam=g.t('cam_logits').clone())
    return g, m, shapes, l2i, t


@pytest.mark.parametrize('raw_cam', [False, True])
@pytest.mark.parametrize('name', ['deform_n6', 'deform_n12_depth', 'deform_edge', 'deform_n24_b2'])
def test_backward_matches_autograd_of_oracle(name, raw_cam):
    """deform_n24_b2: batch 2 - the forward pairs value row b*N + n with the logits of batch (b*N + n) % B
    (deform3d_cross_attn.py:277), so grad_attn_logits of a batch collects rows of both samples.
    raw_cam: the camera weights are the raw logits (Deform3DCrossAttnMP's neighbour pass, multi_point.py:424-430)."""
    from graph_detr4d_amd import ops
    from oracle import torch_oracle as O
    g, m, shapes, l2i, t = _case(name)
    torch.manual_seed(0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    out, _, _ = O.sample_aggregate(leaves['value'], shapes, leaves['ref'], leaves['offsets'], leaves['attn'],
                                   leaves['cam'], l2i, m['pc_range'], m['img_shape'][0], m['img_shape'][1], raw_cam=raw_cam)
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    d = {k: v.cuda() for k, v in t.items()}
    gv, gr, go, ga, gc = ops.cross_attn_bwd(d['value'], shapes, d['ref'], d['offsets'],
                                            d['attn'].view(*d['attn'].shape[:3], 4, 4).contiguous(), d['cam'],
                                            l2i.cuda(), m['pc_range'], m['img_shape'][0], m['img_shape'][1], gout.cuda(),
                                            raw_cam_weights=raw_cam)
    tol = dict(rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(gv.cpu(), leaves['value'].grad, **tol)
    torch.testing.assert_close(ga.cpu().flatten(-2), leaves['attn'].grad, **tol)
    torch.testing.assert_close(gc.cpu(), leaves['cam'].grad, **tol)
    # location gradients are large where the depth is small: compare relative to the tensor scale
    for got, want in ((go.cpu(), leaves['offsets'].grad), (gr.cpu(), leaves['ref'].grad)):
        scale = want.abs().max().clamp(min=1e-6)
        assert ((got - want).abs().max() / scale).item() < 2e-3


def test_backward_full_size_runs_and_is_consistent():
    """BASELINE size: finite gradients; grad_value sums match the forward's linearity
    (sum over value of grad_value * value == sum(out * grad_out))."""
    from graph_detr4d_amd import ops, synthetic
    dev = 'cuda'
    gen = torch.Generator().manual_seed(21)
    q, n = 900, 24
    levels = synthetic.R50_LEVELS
    s = sum(h * w for h, w in levels)
    val = torch.randn(n, s, 8, 32, generator=gen).to(dev)
    l2i = torch.from_numpy(synthetic.camera_rig(4)).unsqueeze(0).to(dev)
    ref = torch.rand(1, q, 3, generator=gen).to(dev)
    offsets = (torch.randn(1, q, 8, 4, 3, generator=gen) * 1.5).to(dev)
    attn = torch.randn(1, q, 8, 4, 4, generator=gen).to(dev)
    cam = torch.randn(1, q, n, generator=gen).to(dev)
    gout = torch.randn(1, q, 256, generator=gen).to(dev)
    out = ops.cross_attn_fwd(val, levels, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600)
    gv, gr, go, ga, gc = ops.cross_attn_bwd(val, levels, ref, offsets, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, gout)
    for t in (gv, gr, go, ga, gc):
        assert torch.isfinite(t).all()
    lhs = (gv.double() * val.double()).sum().item()          # out is linear in value
    rhs = (out.double() * gout.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))
    assert ga.abs().max() > 0 and gc.abs().max() > 0 and go.abs().max() > 0
    # softmax-logit gradients sum to zero per head
    assert ga.sum(dim=(-1, -2)).abs().max().item() < 1e-3


def test_backward_batch3_against_oracle_seeded():
    """B = 3, N = 7 (N not a multiple of B: every sample meets every logit class): all five gradients against autograd
    of the oracle; the logit gradients sum to zero per head."""
    from graph_detr4d_amd import ops, synthetic
    from oracle import torch_oracle as O
    torch.manual_seed(9)
    b, q, n = 3, 40, 7
    levels = [(16, 28), (8, 14), (4, 7), (2, 4)]
    rig = torch.from_numpy(synthetic.camera_rig(2)[:n])
    l2i = rig.unsqueeze(0).expand(b, -1, -1, -1).contiguous()
    t = dict(value=torch.randn(b * n, sum(h * w for h, w in levels), 8, 32), ref=torch.rand(b, q, 3),
             offsets=torch.randn(b, q, 8, 4, 3) * 2.0, attn=torch.randn(b, q, 8, 16), cam=torch.randn(b, q, n))
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    out, _, _ = O.sample_aggregate(leaves['value'], levels, leaves['ref'], leaves['offsets'], leaves['attn'], leaves['cam'], l2i,
                                   synthetic.PC_RANGE, 900, 1600)
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    d = {k: v.cuda() for k, v in t.items()}
    fwd = ops.cross_attn_fwd(d['value'], levels, d['ref'], d['offsets'], d['attn'].view(b, q, 8, 4, 4).contiguous(), d['cam'],
                             l2i.cuda(), synthetic.PC_RANGE, 900, 1600)
    torch.testing.assert_close(fwd.cpu(), out.detach(), rtol=1e-4, atol=1e-4)
    gv, gr, go, ga, gc = ops.cross_attn_bwd(d['value'], levels, d['ref'], d['offsets'], d['attn'].view(b, q, 8, 4, 4).contiguous(),
                                            d['cam'], l2i.cuda(), synthetic.PC_RANGE, 900, 1600, gout.cuda())
    tol = dict(rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(gv.cpu(), leaves['value'].grad, **tol)
    torch.testing.assert_close(ga.cpu().flatten(-2), leaves['attn'].grad, **tol)
    torch.testing.assert_close(gc.cpu(), leaves['cam'].grad, **tol)
    for got, want in ((go.cpu(), leaves['offsets'].grad), (gr.cpu(), leaves['ref'].grad)):
        scale = want.abs().max().clamp(min=1e-6)
        assert ((got - want).abs().max() / scale).item() < 2e-3
    assert ga.sum(dim=(-1, -2)).abs().max().item() < 1e-3
