"""K5 (csrc/match_head.hip): matching-head tail  logits = w2 . act(y + cs) + b2  vs the torch formulation
(TwoLayerdMLP after VideoTextConcat, reference components/DistributionAlign.py:51-98), forward and backward.
This file checks the kernel against the same formula written in torch ops; the check against the ORACLE (oracle.csmm, the
restatement of the reference module) is tests/test_head_gemm_gpu.py (the head as the epilogue of its GEMM, what the models run)
and tests/test_fullsize_gpu.py (the module at the benchmark shape)."""
import pytest
import torch

from shufflingvideosfortsg_amd import functional as TF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid"])
@pytest.mark.parametrize("B,T,H", [(128, 128, 1024), (3, 33, 256), (2, 1, 8), (5, 70, 512),
                                   (37, 100, 384), (700, 8, 256), (300, 16, 128)])    # round 5: the column-sliced backward (whole pairs per workgroup: 1, 3, 1)
def test_match_head_vs_torch(B, T, H, act):
    g = torch.Generator().manual_seed(B + T + H)
    y = torch.randn(B, T, H, generator=g); cs = torch.randn(B, H, generator=g)
    w2 = torch.randn(H, generator=g) / H ** 0.5; b2 = torch.randn(1, generator=g); gl = torch.randn(B, T, generator=g)
    f = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    ref_in = [t.clone().double().requires_grad_(True) for t in (y, cs, w2, b2)]
    ref = f(ref_in[0] + ref_in[1].unsqueeze(1)) @ ref_in[2] + ref_in[3]
    ref.backward(gl.double())
    dev = [t.cuda().requires_grad_(True) for t in (y, cs, w2, b2)]
    out = TF.match_head(*dev, activation=act)
    out.backward(gl.cuda())
    torch.testing.assert_close(out.cpu().double(), ref.detach(), atol=2e-5, rtol=2e-5)
    for got, want, name in zip(dev, ref_in, ("dy", "dcs", "dw2", "db2")):
        scale = max(1.0, want.grad.abs().max().item())
        torch.testing.assert_close(got.grad.cpu().double(), want.grad, atol=1e-5 * scale, rtol=1e-4, msg=lambda s, n=name: f"{n}: {s}")
