"""tsg_linear_fwd (hand-written fp32 MFMA GEMM, the "tsg_gemm_*" row of SURVEY section 8b) vs torch's Linear."""
import pytest
import torch

from shufflingvideosfortsg_amd import functional as TF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(256, 128, 64), (100, 72, 36), (1, 4, 4), (130, 257, 1028), (2048, 1024, 1024), (3, 5, 8)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_hip_matches_float64(shape, bias):
    """Forward and all three gradients; edge tiles (M, N not multiples of 128, K not a multiple of 32); the error vs a
    float64 product must be at the level of torch's own fp32 Linear."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(N, generator=g).cuda().requires_grad_(True) if bias else None
    gy = torch.randn(M, N, generator=g).cuda()
    y = TF.linear_hip(x, w, b)
    y.backward(gy)
    xd, wd = x.detach().double(), w.detach().double()
    ref = xd @ wd.t() + (b.detach().double() if bias else 0.0)
    y32 = torch.nn.functional.linear(x.detach(), w.detach(), b.detach() if bias else None)
    tol = max(4 * (y32.double() - ref).abs().max().item(), 1e-5)
    assert (y.detach().double() - ref).abs().max().item() <= tol
    torch.testing.assert_close(x.grad.double(), gy.double() @ wd, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(w.grad.double(), gy.double().t() @ xd, atol=1e-3, rtol=1e-4)
    if bias:
        torch.testing.assert_close(b.grad.double(), gy.double().sum(0), atol=1e-3, rtol=1e-4)


def test_linear_hip_batched_input_and_errors():
    x = torch.randn(4, 9, 64, device="cuda"); w = torch.randn(32, 64, device="cuda")
    torch.testing.assert_close(TF.linear_hip(x, w), torch.nn.functional.linear(x, w), atol=1e-4, rtol=1e-4)
    with pytest.raises(ValueError):
        TF.linear_hip(torch.randn(4, 6, device="cuda"), torch.randn(3, 6, device="cuda"))      # K % 4 != 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TF.linear_hip(torch.randn(4, 8), torch.randn(3, 8))
