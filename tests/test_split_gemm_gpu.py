"""GPU tests of the split-precision GEMM operand kernel (tsg_split_bf16x3) through the C ABI."""
import pytest
import torch

from shufflingvideosfortsg_amd import functional as TF

pytestmark = pytest.mark.gpu


def test_split_bf16x3_and_split_gemm():
    """tsg_split_bf16x3: planes are exactly (hi,hi,lo)/(hi,lo,hi) with hi = rne_bf16(x), lo = rne_bf16(x-hi), in both
    layouts; the split GEMM's error is at the fp32 GEMM's level (vs float64)."""
    torch.manual_seed(3)
    x = torch.randn(64, 40, device="cuda") * torch.logspace(-6, 6, 40, device="cuda")
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    L = TF.split_bf16x3(x, 1, False); R = TF.split_bf16x3(x, 1, True)
    assert torch.equal(L, torch.cat([hi, hi, lo], 1)) and torch.equal(R, torch.cat([hi, lo, hi], 1))
    L0 = TF.split_bf16x3(x, 0, False); R0 = TF.split_bf16x3(x, 0, True)
    assert torch.equal(L0, torch.cat([hi, hi, lo], 0)) and torch.equal(R0, torch.cat([hi, lo, hi], 0))
    a = torch.randn(512, 1024, device="cuda"); b = torch.randn(1024, 256, device="cuda") * 0.05
    ref = a.double() @ b.double()
    TF.set_gemm_dtype("f32s")
    try:
        for A, B_ in [(a, b), (a.t().contiguous().t(), b), (a, b.t().contiguous().t())]:
            got = TF._mm(A, B_)
            assert got.dtype == torch.float32
            err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-5, err
    finally:
        TF.set_gemm_dtype(None)
