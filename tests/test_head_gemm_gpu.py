"""The heads as the epilogue of their own first-Linear GEMM (tsg_match_head_gemm = K5, tsg_boundary_head_gemm = K3; SURVEY 8f #2
"split-W Linear + ReLU + dot epilogue") against the CPU oracle's un-fused formulation:
  * csmm (DistributionAlign.py:112-118): concat -> Linear(2d, H) -> act -> Linear(H, 1)
  * VideoSentenceConcat + gate + MLP_predictor (CrossModalInteraction.py:44-47, SpanGroundMatchDisc.py:86, SpanPredictor.py:71-85)
through the C ABI (module level, "f32s" mode: the arithmetic of these kernels), outputs at the north star's 1e-4, gradients at the
tolerances of the un-fused K3 / K5 tests; plus: no_grad never writes y, ragged row tiles (T not a multiple of the tile, T < 32), all
three row-tile sizes, bit-reproducibility of the multi-tile ticket reduction, and agreement with the un-fused HIP path at the full
[128 x 128, 1024] matching-head shape."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


@pytest.fixture
def f32s(request):
    from shufflingvideosfortsg_amd import engine
    engine.set_precision("f32s")
    request.addfinalizer(lambda: engine.set_precision(None))


def _csmm_module(d, H, act="relu"):
    from shufflingvideosfortsg_amd.model.components.DistributionAlign import VideoTextSemanticMatch
    return VideoTextSemanticMatch(dict(name="concat", video_dim=d, query_dim=d), dict(name="none", hidden_dim=256, layers=2, dropout=0.0),
                                  dict(name="mlp", activation=act, hidden_dim=H))


@pytest.mark.parametrize("B,T,d,H", [(2, 128, 1024, 1024),     # north-star widths: 4 column tiles per row tile (ticket reduction), 64-row tiles
                                     (16, 100, 256, 256),      # row tiles straddle batch items (T = 100), one column tile
                                     (16, 20, 128, 512),       # T < 32: more than two items per 32-row MFMA tile (per-row lookup of cs)
                                     (4, 64, 64, 256)])        # K = 64: two chunks, the tail-only pipeline
def test_match_head_gemm_vs_oracle(B, T, d, H, f32s):
    from shufflingvideosfortsg_amd import functional as TF
    torch.manual_seed(3)
    m = _csmm_module(d, H)
    g = torch.Generator().manual_seed(B * T + d)
    video = torch.randn(B, T, d, generator=g); sent = torch.randn(B, d, generator=g); gl = torch.randn(B, T, generator=g)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    v0, s0 = video.clone().requires_grad_(True), sent.clone().requires_grad_(True)
    ref = O.csmm(v0, s0, w)
    ref.backward(gl)
    m = m.cuda()
    assert TF.head_gemm_ok(B * T, H, d, T, H), "this shape must take the fused path"
    calls = []
    orig = TF._call
    TF._call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        v1, s1 = video.cuda().requires_grad_(True), sent.cuda().requires_grad_(True)
        out, _ = m(v1, s1, None)
        out.backward(gl.cuda())
        with torch.no_grad():
            out_ng, _ = m(video.cuda(), sent.cuda(), None)
    finally:
        TF._call = orig
    torch.cuda.synchronize()
    assert calls.count("tsg_match_head_gemm") == 2 and "tsg_match_head_fwd" not in calls, calls
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), **TOL)
    assert torch.equal(out_ng, out.detach()), "the no_grad launch (y never written) must give the same logits bit for bit"
    torch.testing.assert_close(v1.grad.cpu(), v0.grad, atol=2e-4, rtol=2e-3)
    torch.testing.assert_close(s1.grad.cpu(), s0.grad, atol=2e-4 * max(1.0, float(s0.grad.abs().max())), rtol=2e-3)
    for k, p in m.named_parameters():
        want = w[k].grad
        torch.testing.assert_close(p.grad.cpu(), want, atol=2e-4 * max(1.0, float(want.abs().max())), rtol=2e-3, msg=lambda s, k=k: f"{k}: {s}")


@pytest.mark.parametrize("act", ["tanh", "sigmoid"])
def test_match_head_gemm_activations(act, f32s):
    """The other two activations of TwoLayerdMLP (DistributionAlign.py:86-92) against the un-fused K5 kernel on the same GEMM output."""
    from shufflingvideosfortsg_amd import functional as TF
    B, T, d, H = 4, 64, 256, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, d, generator=g).cuda(); W = (torch.randn(H, 2 * d, generator=g) / (2 * d) ** 0.5).cuda()
    cs = torch.randn(B, H, generator=g).cuda(); w2 = (torch.randn(H, generator=g) / H ** 0.5).cuda(); b2 = torch.randn(1, generator=g).cuda()
    fused = TF.match_head_gemm(x, W[:, :d], cs, w2, b2, act)
    y = TF.gemm_f32s(x.view(B * T, d), W[:, :d].contiguous()).view(B, T, H) if TF.gemm_f32s_ok(B * T, H, d) else \
        TF._mm(x.view(B * T, d), W[:, :d].t(), "f32s").view(B, T, H)
    plain = TF.match_head(y, cs, w2, b2, act)
    torch.testing.assert_close(fused, plain, atol=2e-5, rtol=2e-5)


def _weights(Dv, Ds, Hm, g):
    p = {}
    for n in ("start", "end"):
        p[f"{n}_mlp_1.weight"] = torch.randn(Hm, Dv + Ds, generator=g) / (Dv + Ds) ** 0.5
        p[f"{n}_mlp_1.bias"] = torch.randn(Hm, generator=g) * 0.1
        p[f"{n}_mlp_2.weight"] = torch.randn(1, Hm, generator=g) / Hm ** 0.5
        p[f"{n}_mlp_2.bias"] = torch.randn(1, generator=g) * 0.1
    return p


@pytest.mark.parametrize("B,T,Dv,Ds,Hm,use_mask,use_gate", [
    (2, 128, 1024, 1024, 256, True, True),     # north-star widths, GMD gate + mask
    (2, 128, 1024, 1024, 256, False, False),   # Baseline: no gate, no mask
    (16, 100, 256, 256, 256, True, True),      # row tiles straddle batch items
    (16, 20, 128, 64, 512, False, True),       # T < 32; Hm = 512: two column tiles per head (ticket reduction per head)
])
def test_boundary_head_gemm_vs_oracle(B, T, Dv, Ds, Hm, use_mask, use_gate, f32s):
    from shufflingvideosfortsg_amd import functional as TF
    from shufflingvideosfortsg_amd.model.components.SpanPredictor import MLP_predictor
    g = torch.Generator().manual_seed(5)
    p = {k: v.requires_grad_(True) for k, v in _weights(Dv, Ds, Hm, g).items()}
    video = torch.randn(B, T, Dv, generator=g, requires_grad=True)
    sent = torch.randn(B, Ds, generator=g, requires_grad=True)
    gate = (torch.randn(B, T, generator=g)).requires_grad_(True) if use_gate else None
    mask = None
    if use_mask:
        n = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None, :] < n[:, None]).int()
    gs, ge = torch.randn(B, T, generator=g), torch.randn(B, T, generator=g)
    x = O.video_sentence_concat(video, sent)
    if use_gate:
        x = gate.unsqueeze(2) * x
    s0, e0 = O.mlp_predictor(x, p, mask)
    (s0 * gs + e0 * ge).sum().backward()

    m = MLP_predictor(Dv + Ds, Hm)
    m.load_state_dict({k: v.detach() for k, v in p.items()})
    m.cuda()
    assert TF.head_gemm_ok(B * T, 2 * Hm, Dv, T, Hm)
    calls = []
    orig = TF._call
    TF._call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        vd, sd = video.detach().cuda().requires_grad_(True), sent.detach().cuda().requires_grad_(True)
        gd = gate.detach().cuda().requires_grad_(True) if use_gate else None
        md = mask.cuda() if mask is not None else None
        s1, e1 = m.forward_split(vd, sd, gd, md)
        (s1 * gs.cuda() + e1 * ge.cuda()).sum().backward()
        with torch.no_grad():
            s2, e2 = m.forward_split(vd.detach(), sd.detach(), gd.detach() if gd is not None else None, md)
    finally:
        TF._call = orig
    torch.cuda.synchronize()
    assert calls.count("tsg_boundary_head_gemm") == 2 and "tsg_boundary_score_fwd" not in calls, calls
    torch.testing.assert_close(s1.detach().cpu(), s0.detach(), **TOL)
    torch.testing.assert_close(e1.detach().cpu(), e0.detach(), **TOL)
    assert torch.equal(s2, s1.detach()) and torch.equal(e2, e1.detach())
    torch.testing.assert_close(s1.sum(1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
    torch.testing.assert_close(vd.grad.cpu(), video.grad, atol=2e-4, rtol=2e-3)
    torch.testing.assert_close(sd.grad.cpu(), sent.grad, atol=2e-4, rtol=2e-3)
    if use_gate:
        torch.testing.assert_close(gd.grad.cpu(), gate.grad, atol=2e-4, rtol=2e-3)
    for k, prm in m.named_parameters():
        torch.testing.assert_close(prm.grad.cpu(), p[k].grad, atol=2e-4 * max(1.0, float(p[k].grad.abs().max())), rtol=2e-3,
                                   msg=lambda s, k=k: f"d{k}: {s}")


def test_match_head_gemm_full_size_vs_unfused_and_reproducible(f32s):
    """[128 items x 128 clips, d = 1024, H = 1024] -- the launch of the train step (original + shuffled video batched): 256-row tiles,
    four column tiles per row tile meeting at a ticket.  Against the un-fused HIP path (own GEMM + K5 kernel) on the same operands, and
    20 launches bit-identical (the last arrival adds the partial rows in tile order, no float atomics)."""
    from shufflingvideosfortsg_amd import functional as TF
    B, T, d, H = 128, 128, 1024, 1024
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, d, generator=g).cuda(); W = (torch.randn(H, 2 * d, generator=g) / (2 * d) ** 0.5).cuda()
    cs = torch.randn(B, H, generator=g).cuda(); w2 = (torch.randn(H, generator=g) / H ** 0.5).cuda(); b2 = torch.randn(1, generator=g).cuda()
    first = TF.match_head_gemm(x, W[:, :d], cs, w2, b2, "relu")
    y = TF.gemm_f32s(x.view(B * T, d), W[:, :d].contiguous()).view(B, T, H)
    plain = TF.match_head(y, cs, w2, b2, "relu")
    torch.testing.assert_close(first, plain, atol=2e-5, rtol=2e-5)
    for _ in range(20):
        assert torch.equal(TF.match_head_gemm(x, W[:, :d], cs, w2, b2, "relu"), first)


def test_gemm_f32s_ld_slices(f32s):
    """tsg_gemm_f32s_ld: operands and output as column slices of wider row-major matrices, against the contiguous call."""
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    M, N, K = 512, 256, 96
    g = torch.Generator().manual_seed(2)
    Xw = torch.randn(M, K + 32, generator=g).cuda(); Ww = torch.randn(N, 2 * K, generator=g).cuda(); bias = torch.randn(N, generator=g).cuda()
    Yw = torch.zeros(M, N + 64, device="cuda")
    x, w = Xw[:, :K].contiguous(), Ww[:, :K].contiguous()
    y = torch.empty(M, N, device="cuda")
    assert lib.tsg_gemm_f32s(ptr(x), ptr(w), ptr(bias), ptr(y), M, N, K, st) == 0, lib.tsg_last_error()
    assert lib.tsg_gemm_f32s_ld(ptr(Xw), K + 32, ptr(Ww), 2 * K, ptr(bias), ptr(Yw), N + 64, M, N, K, st) == 0, lib.tsg_last_error()
    torch.cuda.synchronize()
    assert torch.equal(Yw[:, :N], y) and float(Yw[:, N:].abs().max()) == 0.0
    # (split-precision product of N(0,1) operands: ~2^-16 relative per term; the equality above is the point of this test)
    torch.testing.assert_close(y, x.double().mm(w.double().t()).float() + bias, atol=1e-3, rtol=1e-3)


def _heads_inputs(B, T, Dv, Ds, Hm, H, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()
    return dict(x=r(B, T, Dv), sent=r(B, Ds), Ws=r(Hm, Dv + Ds) * 0.2, bs=r(Hm), We=r(Hm, Dv + Ds) * 0.2, be=r(Hm), w2s=r(1, Hm), b2s=r(1),
                w2e=r(1, Hm), b2e=r(1), gate=torch.sigmoid(r(B, T)), W1=r(H, Dv + Ds) * 0.2, b1=r(H), w2=r(1, H), b2=r(1),
                mask=(torch.rand(B, T, generator=g) > 0.1).cuda())


@pytest.mark.parametrize("B,T,Dv,Ds,Hm", [(4, 128, 1024, 1024, 256), (8, 64, 512, 256, 256)])
def test_boundary_head_params_equals_the_sliced_route(B, T, Dv, Ds, Hm, f32s):
    """_BoundaryHeadFull (parameters in, full-width dW out) against the route it replaces -- column-slice views, torch Linears for the
    sentence half, torch.cat of the small vectors -- which the oracle tests above pin: same probabilities (bit-equal: same kernel, same
    cs up to the small GEMM's arithmetic) and the same gradients for every input and parameter."""
    from shufflingvideosfortsg_amd import functional as TF
    d = _heads_inputs(B, T, Dv, Ds, Hm, 1024, 5)
    names = ("x", "sent", "Ws", "bs", "We", "be", "w2s", "b2s", "w2e", "b2e", "gate")
    gs, ge = torch.randn(B, T, device="cuda"), torch.randn(B, T, device="cuda")

    def run(new):
        p = {k: d[k].clone().requires_grad_(True) for k in names}
        if new:
            ps, pe = TF.boundary_head_params(*(p[k] for k in names), d["mask"])
        else:
            cs = torch.cat([p["sent"] @ p["Ws"][:, Dv:].t(), p["sent"] @ p["We"][:, Dv:].t()], 1)
            ps, pe = TF.boundary_head_gemm(p["x"], p["Ws"][:, :Dv], p["We"][:, :Dv], cs, torch.cat([p["bs"], p["be"]]),
                                           torch.cat([p["w2s"].reshape(-1), p["w2e"].reshape(-1)]), torch.cat([p["b2s"], p["b2e"]]),
                                           p["gate"], d["mask"])
        ((ps * gs).sum() + (pe * ge).sum()).backward()
        return ps.detach(), pe.detach(), {k: p[k].grad for k in names}

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in names:
        scale = max(1e-6, float(b[2][k].abs().max()))
        torch.testing.assert_close(a[2][k], b[2][k], atol=2e-5 * scale, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("B,T,Dv,Dq,H", [(8, 128, 1024, 1024, 1024), (16, 64, 512, 256, 512)])
def test_match_head_params_equals_the_sliced_route(B, T, Dv, Dq, H, f32s):
    from shufflingvideosfortsg_amd import functional as TF
    d = _heads_inputs(B, T, Dv, Dq, 256, H, 6)
    names = ("x", "sent", "W1", "b1", "w2", "b2")
    gl = torch.randn(B, T, device="cuda")

    def run(new):
        p = {k: d[k].clone().requires_grad_(True) for k in names}
        if new:
            out = TF.match_head_params(*(p[k] for k in names), "relu")
        else:
            cs = torch.addmm(p["b1"], p["sent"], p["W1"][:, Dv:].t())
            out = TF.match_head_gemm(p["x"], p["W1"][:, :Dv], cs, p["w2"], p["b2"], "relu")
        (out * gl).sum().backward()
        return out.detach(), {k: p[k].grad for k in names}

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    for k in names:
        scale = max(1e-6, float(b[1][k].abs().max()))
        torch.testing.assert_close(a[1][k], b[1][k], atol=2e-5 * scale, rtol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
