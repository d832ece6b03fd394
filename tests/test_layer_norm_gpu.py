"""LayerNorm of the query-aware video encoder (reference components/VideoEncoder.py:96,112: `nn.LayerNorm(d)`, eps 1e-5) on the hand-written
kernels (tsg_layer_norm_fwd / _bwd) against torch.nn.functional.layer_norm in float64 -- the arithmetic the reference's module and the
oracle's `query_aware_encoder` run (oracle/tsg_oracle.py: F.layer_norm) -- forward, dx, dgamma, dbeta; ragged widths, many rows, bf16."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,d", [(7, 24), (16384, 1024), (1000, 512), (333, 260), (64, 2048), (5, 1536)])
def test_layer_norm_vs_float64(rows, d):
    from shufflingvideosfortsg_amd import functional as TF
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).requires_grad_(True); beta = (0.1 * torch.randn(d, generator=g)).requires_grad_(True)
    gy = torch.randn(rows, d, generator=g)
    xd, gd, bd = x.detach().double().requires_grad_(True), gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    y0 = torch.nn.functional.layer_norm(xd, (d,), gd, bd, 1e-5)
    y0.backward(gy.double())
    xc, gc, bc = (t.detach().cuda().requires_grad_(True) for t in (x, gamma, beta))
    y1 = TF.layer_norm(xc.view(rows, 1, d), gc, bc, 1e-5).view(rows, d)
    y1.backward(gy.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(y1.detach().cpu().double(), y0.detach(), atol=2e-6, rtol=2e-6)
    torch.testing.assert_close(xc.grad.cpu().double(), xd.grad, atol=5e-6, rtol=1e-5)
    scale = max(1.0, float(gd.grad.abs().max()))
    torch.testing.assert_close(gc.grad.cpu().double(), gd.grad, atol=1e-5 * scale, rtol=1e-5)
    torch.testing.assert_close(bc.grad.cpu().double(), bd.grad, atol=1e-5 * max(1.0, float(bd.grad.abs().max())), rtol=1e-5)
    # fixed summation order: bit-identical parameter gradients run to run
    xc2, gc2, bc2 = (t.detach().cuda().requires_grad_(True) for t in (x, gamma, beta))
    TF.layer_norm(xc2, gc2, bc2, 1e-5).backward(gy.cuda())
    assert torch.equal(gc2.grad, gc.grad) and torch.equal(bc2.grad, bc.grad) and torch.equal(xc2.grad, xc.grad)


def test_layer_norm_bf16_storage():
    from shufflingvideosfortsg_amd import functional as TF
    rows, d = 4096, 1024
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, d, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).requires_grad_(True); beta = (0.1 * torch.randn(d, generator=g)).requires_grad_(True)
    gy = torch.randn(rows, d, generator=g).to(torch.bfloat16).float()
    y0 = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    y0.backward(gy)
    xc = x.detach().to(torch.bfloat16).cuda().requires_grad_(True)
    gc, bc = gamma.detach().cuda().requires_grad_(True), beta.detach().cuda().requires_grad_(True)
    y1 = TF.layer_norm(xc, gc, bc, 1e-5)
    assert y1.dtype == torch.bfloat16
    y1.backward(gy.to(torch.bfloat16).cuda())
    assert xc.grad.dtype == torch.bfloat16 and gc.grad.dtype == torch.float32
    torch.testing.assert_close(y1.float().cpu(), y0.detach(), atol=3e-2, rtol=1e-2)          # one bf16 rounding of the output
    torch.testing.assert_close(xc.grad.float().cpu(), x.grad, atol=1e-2 * float(x.grad.abs().max()), rtol=1e-2)
    torch.testing.assert_close(gc.grad.cpu(), gamma.grad, atol=2e-3 * float(gamma.grad.abs().max()), rtol=2e-3)
    torch.testing.assert_close(bc.grad.cpu(), beta.grad, atol=2e-3 * float(beta.grad.abs().max()), rtol=2e-3)


def test_layer_norm_rejects_wide_rows():
    from shufflingvideosfortsg_amd import functional as TF
    x = torch.randn(4, 4096, device="cuda")
    assert not TF.layer_norm_ok(x)
    with pytest.raises(RuntimeError, match="multiple of 4, <= 2048"):
        TF.layer_norm(x, torch.ones(4096, device="cuda"), torch.zeros(4096, device="cuda"))
