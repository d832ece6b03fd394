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
    (1, 40, 33, 640, 640, 2, False),    # head width 320 (> 256: VALU kernels)
    (2, 128, 128, 2048, 2048, 8, False),  # head width 256 (Self_Attention_predictor at d=1024): wide MFMA kernels, Tk = 128
    (1, 96, 512, 512, 512, 2, False),   # head width 256, Tk = 512 (config 4), ragged query tile
    (1, 70, 70, 384, 256, 2, True),     # head widths 192 / 128 on the wide kernels, causal, ragged
    (2, 33, 20, 1024, 1024, 4, False),  # head width 256, cross attention (Tk = 20)
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


@pytest.mark.parametrize("heads", [1, 4])
def test_attention_dropout_mask_forward_backward(heads):
    """Attention dropout inside K2 (reference: out = dropout(softmax(A)) V, attention.py:53-54).  With V = one identity
    block per head the output IS dropout(S), which exposes the kernel's keep mask: the kept fraction is 1-p, kept
    entries are S/(1-p), the MFMA and the VALU kernels draw the same mask, a different offset draws a different one,
    and the backward equals autograd through the same masked expression."""
    from shufflingvideosfortsg_amd import functional as TF
    torch.manual_seed(0)
    B, Tq, Tk, dh, p = 3, 40, 32, 16, 0.3
    dk, dv = heads * dh, heads * Tk
    Q = torch.randn(B, Tq, dk, device="cuda"); K = torch.randn(B, Tk, dk, device="cuda")
    Vi = torch.zeros(B, Tk, dv, device="cuda")
    for hd in range(heads):
        Vi[:, :, hd * Tk:(hd + 1) * Tk] = torch.eye(Tk, device="cuda")
    scale = dk ** 0.5
    S0 = TF._MHA.apply(Q, K, Vi, heads, scale, False, False, 0.0, 0, 0)[0]            # un-dropped softmax per head
    Sd = TF._MHA.apply(Q, K, Vi, heads, scale, False, False, p, 1234, 77)[0]
    M = Sd != 0
    kept = M.float().mean().item()
    n = M.numel()
    assert abs(kept - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5 + 1e-3, kept
    torch.testing.assert_close(Sd, torch.where(M, S0 / (1 - p), torch.zeros_like(S0)), atol=1e-6, rtol=1e-5)
    Sd_valu = TF._MHA.apply(Q, K, Vi, heads, scale, False, True, p, 1234, 77)[0]      # return_maps -> VALU kernel
    torch.testing.assert_close(Sd_valu, Sd, atol=1e-6, rtol=1e-5)
    assert not torch.equal(TF._MHA.apply(Q, K, Vi, heads, scale, False, False, p, 1234, 78)[0] != 0, M)
    assert torch.equal(TF._MHA.apply(Q, K, Vi, heads, scale, False, False, p, 1234, 77)[0] != 0, M)
    # backward with a real V, against autograd through softmax * mask / (1-p)
    V = torch.randn(B, Tk, dv, device="cuda")
    for maps in (False, True):
        q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
        O = TF._MHA.apply(q, k, v, heads, scale, False, maps, p, 1234, 77)[0]
        g = torch.randn_like(O)
        O.backward(g)
        qr, kr, vr = (x.clone().requires_grad_(True) for x in (Q, K, V))
        outs = []
        for hd in range(heads):
            a = qr[..., hd * dh:(hd + 1) * dh] @ kr[..., hd * dh:(hd + 1) * dh].transpose(1, 2) / scale
            sm = torch.softmax(a, -1) * M[..., hd * Tk:(hd + 1) * Tk].float() / (1 - p)
            outs.append(sm @ vr[..., hd * Tk:(hd + 1) * Tk])
        Oref = torch.cat(outs, -1)
        Oref.backward(g)
        torch.testing.assert_close(O.detach(), Oref.detach(), atol=1e-4, rtol=1e-4)
        for got, want, name in ((q, qr, "dQ"), (k, kr, "dK"), (v, vr, "dV")):
            torch.testing.assert_close(got.grad, want.grad, atol=2e-4, rtol=2e-3, msg=lambda m, n=name: f"{n}: {m}")


def test_multihead_module_dropout_modes():
    """MultiHead with drop_ratio > 0: training mode drops (fresh mask per call, reproducible under manual_seed),
    eval mode does not."""
    from shufflingvideosfortsg_amd.model.networks.attention import MultiHead
    torch.manual_seed(1)
    m = MultiHead(64, 64, 4, 0.5).cuda()
    x = torch.randn(2, 24, 64, device="cuda", requires_grad=True)
    m.eval()
    e1, e2 = m(x, x, x), m(x, x, x)
    assert torch.equal(e1, e2)
    m.train()
    torch.manual_seed(5); t1 = m(x, x, x)
    t2 = m(x, x, x)
    torch.manual_seed(5); t3 = m(x, x, x)
    assert not torch.equal(t1, t2) and torch.equal(t1, t3) and not torch.allclose(t1, e1)
    t1.sum().backward()
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_attention_dropout_under_graph_capture():
    """K2's in-kernel dropout inside a captured HIP graph: (seed, offset) live in device memory (tsg_mha_fwd_rng / _bwd_rng) and
    the captured increment of the offset gives every REPLAY a fresh mask; forward and backward of one replay share it, and the
    result equals the host-offset entry points called with the same (seed, offset)."""
    from shufflingvideosfortsg_amd import functional as F
    torch.manual_seed(5)
    B, T, d, h, p = 2, 64, 256, 4, 0.3
    Q = torch.randn(B, T, d, device="cuda", requires_grad=True); K = torch.randn(B, T, d, device="cuda", requires_grad=True)
    V = torch.randn(B, T, d, device="cuda", requires_grad=True); gO = torch.randn(B, T, d, device="cuda")
    st = F.mha_graph_rng(Q.device)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                                     # warm-up (eager path: host offsets)
            o = F.mha(Q, K, V, h, math.sqrt(d), p_drop=p); o.backward(gO)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    Q.grad = K.grad = V.grad = None
    with torch.cuda.graph(g, stream=side):
        out = F.mha(Q, K, V, h, math.sqrt(d), p_drop=p)
        out.backward(gO)
    res = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize()
        seed, off = (int(x) for x in st.tolist())
        ref = F._MHA.apply(Q.detach(), K.detach(), V.detach(), h, math.sqrt(d), False, False, p, seed, off, None)[0]
        assert torch.equal(out, ref), "replayed output differs from the host-offset kernel at the same (seed, offset)"
        # backward of the replay used the same mask: compare with an eager backward at that offset
        Qe, Ke, Ve = (t.detach().clone().requires_grad_(True) for t in (Q, K, V))
        F._MHA.apply(Qe, Ke, Ve, h, math.sqrt(d), False, False, p, seed, off, None)[0].backward(gO)
        assert torch.equal(Q.grad, Qe.grad) and torch.equal(V.grad, Ve.grad)
        res.append(out.clone())
    assert not torch.equal(res[0], res[1]) and not torch.equal(res[1], res[2]), "replays repeated the dropout mask"
    zeros = (res[0] == 0).float().mean().item()
    assert zeros < 0.05                                        # dropout acts on the softmax, not on the output elements


@pytest.mark.parametrize("B,Tq,Tk,d,h,causal,p", [
    (2, 128, 128, 1024, 8, False, 0.0),   # temporal self-attention at the north-star width: head width 128
    (2, 128, 20, 1024, 8, False, 0.0),    # cross attention, Tk = 20: one key tile -> the fused single-tile backward
    (1, 300, 20, 512, 8, False, 0.0),     # one key tile, three query blocks (dK / dV added with atomics), head width 64
    (2, 100, 32, 768, 8, False, 0.2),     # full key tile, head width 96, ragged query block, dropout
    (3, 20, 20, 256, 8, True, 0.0),       # tiny causal self-attention inside one tile, head width 32
    (1, 70, 70, 256, 2, True, 0.0),       # ragged tiles, causal
    (2, 33, 45, 256, 8, False, 0.0),      # head width 32
    (1, 300, 257, 192, 2, False, 0.0),    # head width 96, several 128-row blocks, ragged
    (2, 64, 64, 512, 8, False, 0.3),      # head width 64, attention dropout (mask regenerated from the counters)
    (1, 512, 512, 1024, 8, True, 0.1),    # config-4 length, causal + dropout
    (2, 128, 128, 2048, 8, False, 0.0),   # head width 256 (Self_Attention_predictor at d = 1024): split forward, wide split backward (channel halves per wave pair)
    (1, 200, 150, 1280, 8, False, 0.25),  # head width 160 (second half: one tile), ragged blocks, dropout
    (1, 70, 70, 448, 2, True, 0.0),       # head width 224, causal, ragged
    (1, 70, 45, 384, 2, True, 0.1),       # head width 192, ragged, causal needs Tq == Tk -> see below
    (2, 100, 20, 1280, 8, False, 0.0),    # head width 160, one key tile (wide backward with an idle key group)
    (1, 300, 100, 256, 2, False, 0.1),    # one ragged key block no wider than the head, three query blocks, dropout: dS written once, dQ = dS K (round 6)
    (2, 40, 96, 768, 8, False, 0.0),      # the same path at head width 96 (three key tiles of the head's three)
])
def test_mha_split_precision(B, Tq, Tk, d, h, causal, p):
    """dtype TSG_F32S (the 'f32s' GEMM mode): the attention products as bf16 hi/lo products on the MFMA -- one forward kernel with an
    online softmax, two backward kernels.  Same outputs and gradients as the exact-fp32 kernels to fp32-GEMM-level error, and --
    without dropout -- as float64 autograd of the formula."""
    from shufflingvideosfortsg_amd import functional as F
    if causal and Tq != Tk:
        Tk = Tq                                                  # the reference's causal triangle is [Tk, Tk]
    g = torch.Generator().manual_seed(Tq + Tk)
    Q, K, V = (torch.randn(B, n, d, generator=g).cuda() for n in (Tq, Tk, Tk))
    gO = torch.randn(B, Tq, d, generator=g).cuda()

    def grads(mode):
        F.set_gemm_dtype(mode)
        try:
            q, k, v = (t.clone().requires_grad_(True) for t in (Q, K, V))
            torch.manual_seed(5)                                 # the same dropout (seed, offset) in both runs
            o = F.mha(q, k, v, h, math.sqrt(d), causal, p_drop=p)
            o.backward(gO)
            return o.detach(), [t.grad for t in (q, k, v)]
        finally:
            F.set_gemm_dtype(None)

    o_s, g_s = grads("f32s")
    o_r, g_r = grads("f32s")                                     # every output has one writer (or an ordered fold): repeatable bit for bit,
    assert torch.equal(o_s, o_r) and torch.equal(g_s[0], g_r[0])  # except dK / dV of the single-key-tile kernel with several query
    if not (Tk <= 32 and Tq > 128):                               # blocks, which are added with float atomics
        assert torch.equal(g_s[1], g_r[1]) and torch.equal(g_s[2], g_r[2])
    o_e, g_e = grads(None)
    torch.testing.assert_close(o_s, o_e, atol=2e-5 * float(o_e.abs().max()), rtol=1e-4)        # forward: split vs exact kernels
    for a, b_, name in zip(g_s, g_e, "QKV"):
        scale = float(b_.abs().max())
        torch.testing.assert_close(a, b_, atol=2e-5 * scale + 1e-7, rtol=1e-4, msg=lambda m, n=name: f"d{n} vs fp32 kernels: {m}")
    if p == 0.0:
        q, k, v = (t.double().requires_grad_(True) for t in (Q, K, V))
        dh = d // h
        qh, kh, vh = (t.view(B, -1, h, dh).transpose(1, 2) for t in (q, k, v))
        a = qh @ kh.transpose(-1, -2)
        if causal:
            a = a - 1e10 * torch.triu(torch.ones(Tk, Tk, device="cuda", dtype=torch.float64), 1)
        o = (torch.softmax(a / math.sqrt(d), -1) @ vh).transpose(1, 2).reshape(B, Tq, d)
        o.backward(gO.double())
        assert float((o_s.double() - o.detach()).abs().max()) < 2e-5 * float(o.detach().abs().max())
        for a_, t, name in zip(g_s, (q, k, v), "QKV"):
            scale = float(t.grad.abs().max())
            assert float((a_.double() - t.grad).abs().max()) < 2e-5 * scale + 1e-7, name


def test_split_precision_attention_in_a_captured_graph():
    """The split-precision forward / backward kernels inside a HIP graph (the f32s train step can be graph-replayed with a
    self-attention head): same results as the eager launches, with dropout drawing a fresh mask per replay."""
    from shufflingvideosfortsg_amd import functional as F
    torch.manual_seed(0)
    B, T, d, h = 2, 96, 256, 2                                   # head width 128
    q, k, v = (torch.randn(B, T, d, device="cuda", requires_grad=True) for _ in range(3))
    gO = torch.randn(B, T, d, device="cuda")
    F.set_gemm_dtype("f32s")
    try:
        def run(p):
            for t in (q, k, v):
                t.grad = None
            o = F.mha(q, k, v, h, math.sqrt(d), False, p_drop=p)
            o.backward(gO)
            return o.detach().clone(), q.grad.clone()
        o_e, g_e = run(0.0)
        F.mha_graph_rng("cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(0.0); run(0.2)                                   # warm-up (kernel attributes, allocator)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                o0, g0 = run(0.0)
                o1, g1 = run(0.2)
        torch.cuda.current_stream().wait_stream(s)
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(o0, o_e) and torch.equal(g0, g_e)
        a = o1.clone()
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(o0, o_e) and not torch.equal(o1, a)   # the captured offset increment gives every replay a new mask
    finally:
        F.set_gemm_dtype(None)
