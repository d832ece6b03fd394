"""K2 parity on the GPU: tsg_mha_{fwd,bwd} vs the CPU oracle's MultiHead core."""
import math

import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,Tq,Tk,d,dv,h,causal", [
    (2, 9, 5, 32, 32, 4, False),        # cross, tiny
    (2, 9, 9, 32, 32, 8, False),        # self, head width 4
    (2, 7, 7, 16, 16, 2, True),         # causal
    (1, 4, 6, 8, 8, 1, False),          # single head
    (2, 32, 15, 512, 512, 8, False),    # config-0 shape, cross
    (2, 128, 20, 1024, 1024, 8, False), # north-star shape, cross (B reduced)
    (1, 128, 128, 1024, 1024, 8, False),  # temporal self-attention over 128 clips
    (1, 70, 70, 256, 128, 2, True),     # ragged T, d_value != d_key, head width 128/64, causal
    (1, 40, 33, 640, 640, 2, False),    # head width 320 (> one 128-channel chunk)
])
def test_mha_parity(B, Tq, Tk, d, dv, h, causal):
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(7)
    Q = torch.randn(B, Tq, d, generator=g, requires_grad=True)
    K = torch.randn(B, Tk, d, generator=g, requires_grad=True)
    V = torch.randn(B, Tk, dv, generator=g, requires_grad=True)
    gO = torch.randn(B, Tq, dv, generator=g)
    o0, A0, S0 = O.mha_core(Q, K, V, h, d, causal)       # scale = sqrt(d_model): quirk F2
    o0.backward(gO)
    ref = [t.grad.clone() for t in (Q, K, V)]
    Qd, Kd, Vd = (t.detach().cuda().requires_grad_(True) for t in (Q, K, V))
    o1, A1, S1 = F.mha(Qd, Kd, Vd, h, math.sqrt(d), causal, return_maps=True)
    o1.backward(gO.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(o1.detach().cpu(), o0.detach(), **TOL)
    torch.testing.assert_close(S1.cpu(), S0.detach(), **TOL)
    torch.testing.assert_close(A1.cpu(), A0.detach(), atol=1e-3, rtol=1e-5)   # causal entries ~ -1e10/sqrt(d)
    for got, want, name in zip((Qd, Kd, Vd), ref, "QKV"):
        torch.testing.assert_close(got.grad.cpu(), want, atol=2e-4, rtol=1e-3, msg=lambda m, n=name: f"d{n}: {m}")
    # without the side outputs the forward is the same
    o2 = F.mha(Qd.detach(), Kd.detach(), Vd.detach(), h, math.sqrt(d), causal)
    torch.testing.assert_close(o2.cpu(), o0.detach(), **TOL)


def test_mha_scale_is_full_width():
    """F2: 1/sqrt(d_head) must NOT match the reference."""
    from shufflingvideosfortsg_amd import functional as F
    g = torch.Generator().manual_seed(1)
    Q, K, V = (torch.randn(1, 8, 64, generator=g) for _ in range(3))
    ref, _, _ = O.mha_core(Q, K, V, 4, 64)
    good = F.mha(Q.cuda(), K.cuda(), V.cuda(), 4, math.sqrt(64)).cpu()
    bad = F.mha(Q.cuda(), K.cuda(), V.cuda(), 4, math.sqrt(16)).cpu()
    torch.testing.assert_close(good, ref, **TOL)
    assert (bad - ref).abs().max() > 1e-2


@pytest.mark.parametrize("tag", ["cross", "self", "causal", "onehead"])
def test_mha_golden(golden, tag):
    """Reference MultiHead captured in tests/golden/mha_*.npz: projections by torch, core by the kernel."""
    from shufflingvideosfortsg_amd import functional as F
    lin = torch.nn.functional.linear
    g = golden("mha_" + tag)
    w = {k: v.cuda() for k, v in g.weights.items()}
    h, causal = int(g.a["n_heads"]), bool(g.a["causal"])
    q = g.t("q").cuda()
    k = q if tag in ("self", "causal") else g.t("k").cuda()
    v = q if tag in ("self", "causal") else g.t("v").cuda()
    d = q.shape[-1]
    o, A, S = F.mha(lin(q, w["wq.weight"]), lin(k, w["wk.weight"]), lin(v, w["wv.weight"]), h, math.sqrt(d), causal, True)
    torch.testing.assert_close(lin(o, w["wo.weight"]).cpu(), g.t("out"), **TOL)
    torch.testing.assert_close(S.cpu(), g.t("A_softmax"), **TOL)
    torch.testing.assert_close(A.cpu(), g.t("A"), atol=1e-3, rtol=1e-5)
