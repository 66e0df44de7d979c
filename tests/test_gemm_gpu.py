"""gd4d_gemm_bf16x3_fwd / gd4d_split_bf16_fwd against fp64.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_is_exact_to_16_bits():
    from graph_detr4d_amd import ops
    torch.manual_seed(0)
    w = (torch.randn(300, 77) * torch.logspace(-3, 3, 77)).cuda()
    hi, lo = ops.split_bf16_fwd(w)
    assert hi.dtype == torch.bfloat16 and lo.shape == w.shape
    err = (w - (hi.float() + lo.float())).abs()
    assert bool((err <= w.abs() * 2.0 ** -16 + 1e-38).all())


@pytest.mark.parametrize('m,k,n,relu', [(1000, 192, 1024, True), (777, 1024, 256, False), (128, 32, 256, False),
                                        (5, 256, 256, True), (4097, 64, 256, False)])
def test_gemm_bf16x3_matches_fp64(m, k, n, relu):
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    a, w, b = torch.randn(m, k), torch.randn(n, k) * 0.1, torch.randn(n)
    w[3, 2] = 2.5
    hi, lo = ops.split_bf16_fwd(w.cuda())
    got = ops.gemm_bf16x3_fwd(a.cuda(), hi, lo, b.cuda(), relu=relu).cpu()
    ref = a.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.relu()
    scale = (a.double().abs() @ w.double().abs().t()).max().item()        # magnitude of the products summed
    assert (got.double() - ref).abs().max().item() < 4e-5 * max(1.0, scale / 8)
    assert got.shape == (m, n)


def test_gemm_errors():
    from graph_detr4d_amd import ops
    from graph_detr4d_amd._lib import Gd4dError
    a = torch.randn(10, 48).cuda()
    hi, lo = ops.split_bf16_fwd(torch.randn(256, 48).cuda())
    with pytest.raises(Gd4dError):                       # K % 32
        ops.gemm_bf16x3_fwd(a, hi, lo)
    a = torch.randn(10, 64).cuda()
    hi, lo = ops.split_bf16_fwd(torch.randn(128, 64).cuda())
    with pytest.raises(Gd4dError):                       # N % 256
        ops.gemm_bf16x3_fwd(a, hi, lo)
