"""gd4d_value_proj_fwd (split-bf16 x3 MFMA, NCHW -> channels-last) against an fp64 reference GEMM
and against the fp32 torch Linear the reference runs.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(feats, w, b):
    flat = torch.cat([f.reshape(-1, f.shape[-3], f.shape[-2] * f.shape[-1]) for f in feats], 2)
    return torch.matmul(flat.transpose(1, 2).double(), w.double().t()) + b.double()


@pytest.mark.parametrize('levels', [[(16, 28), (8, 14), (4, 7), (2, 4)],     # ragged tails at every level
                                    [(5, 13)], [(1, 1), (1, 3)], [(8, 8), (8, 8), (3, 3)]])
@pytest.mark.parametrize('out_dtype', [torch.float32, torch.bfloat16])
def test_value_proj_matches_fp64(levels, out_dtype):
    from graph_detr4d_amd import ops
    torch.manual_seed(3)
    r = 5
    feats = [torch.randn(r, 256, h, w) for h, w in levels]
    w = torch.randn(256, 256) * 0.06
    w[3, 7] = 1.0                      # asymmetric landmarks: catches transposed operands / outputs
    w[200, 1] = -2.0
    b = torch.randn(256)
    got = ops.value_proj_fwd([f.cuda() for f in feats], w.cuda(), b.cuda(), out_dtype).float().cpu()
    ref = _ref(feats, w, b)
    if out_dtype == torch.float32:
        # fp32-class: split-bf16 x3 keeps ~2^-17 relative per product
        assert (got.double() - ref).abs().max().item() < 5e-5
        f32 = torch.nn.functional.linear(torch.cat([f.reshape(r, 256, -1) for f in feats], 2).transpose(1, 2), w, b)
        assert (got - f32).abs().max().item() < 5e-5
    else:
        torch.testing.assert_close(got.double(), ref, rtol=1e-2, atol=1e-2)
        assert (got.double() - ref.float().bfloat16().double()).abs().max().item() < 0.04


def test_value_proj_identity_weight_is_a_transpose():
    """W = I, bias = 0 -> out[r, pix, c] == in[r, c, pix] to split precision; layout check."""
    from graph_detr4d_amd import ops
    torch.manual_seed(4)
    f = torch.randn(3, 256, 9, 11)
    got = ops.value_proj_fwd([f.cuda()], torch.eye(256).cuda(), None).cpu()
    want = f.reshape(3, 256, 99).transpose(1, 2)
    # x = hi + lo keeps 16 significant bits: |err| <= 2^-16 |x|
    assert ((got - want).abs() <= want.abs() * 2.0 ** -16 + 1e-7).all()
    assert got.shape == (3, 99, 256)


def test_value_proj_full_size_properties():
    """BASELINE size (24 cameras, 116x200..15x25): linearity in the input and agreement with an
    fp32 GEMM on a random subset of pixels."""
    from graph_detr4d_amd import ops, synthetic
    torch.manual_seed(6)
    dev = 'cuda'
    feats = [torch.randn(24, 256, h, w, device=dev) for h, w in synthetic.R50_LEVELS]
    w = (torch.randn(256, 256, device=dev) * 0.06)
    b = torch.randn(256, device=dev)
    o1 = ops.value_proj_fwd(feats, w, b)
    o2 = ops.value_proj_fwd([2.0 * f for f in feats], w, None)
    torch.testing.assert_close(o2 + b, 2.0 * (o1 - b) + b, rtol=1e-4, atol=1e-4)
    idx = torch.randint(0, o1.shape[1], (500,), device=dev)
    flat = torch.cat([f.reshape(24, 256, -1) for f in feats], 2)[:, :, idx]      # (24, 256, 500)
    ref = torch.matmul(flat.transpose(1, 2).double(), w.double().t()) + b.double()
    assert (o1[:, idx].double() - ref).abs().max().item() < 5e-5


def test_multi_layer_projection_is_bit_identical_to_single_calls():
    """gd4d_value_proj_multi_fwd (6 layers, one launch) == 6 x gd4d_value_proj_fwd, bit for bit."""
    from graph_detr4d_amd import ops
    torch.manual_seed(8)
    feats = [torch.randn(7, 256, h, w).cuda() for h, w in [(20, 36), (10, 18), (5, 9), (3, 5)]]
    ws = [(torch.randn(256, 256) * 0.06).cuda() for _ in range(6)]
    bs = [torch.randn(256).cuda() if i % 2 == 0 else None for i in range(6)]
    multi = ops.value_proj_multi_fwd(feats, ws, bs)
    for w, b, m in zip(ws, bs, multi):
        assert torch.equal(m, ops.value_proj_fwd(feats, w, b))


@pytest.mark.parametrize('out_dtype', [torch.float32, torch.bfloat16])
def test_head_major_layout_is_a_permutation_of_pixel_major(out_dtype):
    """GD4D_LAYOUT_HEAD_MAJOR output (R, Hh, S, Dh) == pixel-major (R, S, Hh, Dh) permuted, bit for bit."""
    from graph_detr4d_amd import ops
    torch.manual_seed(10)
    feats = [torch.randn(5, 256, h, w).cuda() for h, w in [(20, 36), (10, 18), (5, 9), (3, 5)]]
    w, b = (torch.randn(256, 256) * 0.06).cuda(), torch.randn(256).cuda()
    pm = ops.value_proj_fwd(feats, w, b, out_dtype)
    hm = ops.value_proj_fwd(feats, w, b, out_dtype, head_major=True)
    assert hm.shape == (5, 8, pm.shape[1], 32)
    assert torch.equal(hm, pm.view(5, -1, 8, 32).permute(0, 2, 1, 3))
    hm6 = ops.value_proj_multi_fwd(feats, [w] * 3, [b] * 3, out_dtype, head_major=True)
    assert all(torch.equal(x, hm) for x in hm6)


@pytest.mark.parametrize('head_major', [False, True])
def test_bf16_math_mode(head_major):
    """GD4D_VP_PRECISION_BF16: one bf16 product per MAC, fp32 accumulate, bf16 output - equals an fp64 GEMM
    of the bf16-rounded operands up to accumulation order and the final bf16 rounding."""
    from graph_detr4d_amd import ops
    torch.manual_seed(12)
    feats = [torch.randn(4, 256, h, w) for h, w in [(16, 28), (8, 14), (4, 7), (2, 4)]]
    w, b = torch.randn(256, 256) * 0.06, torch.randn(256)
    got = ops.value_proj_fwd([f.cuda() for f in feats], w.cuda(), b.cuda(), torch.bfloat16, head_major=head_major,
                             bf16_math=True).float().cpu()
    if head_major:
        got = got.permute(0, 2, 1, 3).reshape(4, -1, 256)
    flat = torch.cat([f.reshape(4, 256, -1) for f in feats], 2).bfloat16().double()
    ref = torch.matmul(flat.transpose(1, 2), w.bfloat16().double().t()) + b.double()
    assert (got.double() - ref).abs().max().item() < 0.03          # bf16 output rounding at |v| <= 4
    assert (got.double() - ref).abs().mean().item() < 3e-3


def test_value_proj_repeatable_under_load():
    """Race screen for the pipelined kernel (LDS-DMA + counted vmcnt across a raw barrier): 12 launches of
    the full-size problem with other traffic in between must be bit-identical."""
    from graph_detr4d_amd import ops, synthetic
    torch.manual_seed(13)
    dev = 'cuda'
    feats = [torch.randn(24, 256, h, w, device=dev) for h, w in synthetic.R50_LEVELS]
    w, b = torch.randn(256, 256, device=dev) * 0.06, torch.randn(256, device=dev)
    ref = ops.value_proj_fwd(feats, w, b).clone()
    junk = torch.empty(64 << 20, device=dev)
    for i in range(12):
        junk.normal_()                                     # unrelated HBM traffic / cache pollution
        got = ops.value_proj_fwd(feats, w, b)
        assert torch.equal(got, ref), f'run {i} differs'
