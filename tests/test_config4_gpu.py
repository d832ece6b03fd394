"""BASELINE config 4 readiness: ActivityNet-CD shape T_clip=512, T_word=25, d=1024 (cfgs/anet_cd_i3d.yml:17-25), B=128 per GPU
(= 256 batched rows through the shared-weight video encoder).  Parity of every kernel on the path at T=512 with the batch
reduced for the CPU oracle, and the persistent LSTM's row chunking at 256 rows."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)
T4, N4, D4 = 512, 25, 1024


def test_k1_t512_vs_oracle():
    from shufflingvideosfortsg_amd import functional as F
    B = 2
    g = torch.Generator().manual_seed(4)
    a = torch.randn(B, T4, D4, generator=g).requires_grad_(True); s = torch.randn(B, N4, D4, generator=g).requires_grad_(True)
    w = (torch.randn(D4, generator=g) / D4 ** 0.5).requires_grad_(True); sent = torch.randn(B, N4, D4, generator=g).requires_grad_(True)
    gC = torch.randn(B, T4, D4, generator=g)
    C0, P0 = O.scdm_core(a, s, w, sent)
    C0.backward(gC)
    dev = [x.detach().cuda().requires_grad_(True) for x in (a, s, w, sent)]
    C1, P1 = F.scdm_attn(*dev, return_p=True)
    C1.backward(gC.cuda())
    torch.testing.assert_close(C1.detach().cpu(), C0.detach(), **TOL)
    torch.testing.assert_close(P1.cpu(), P0.detach(), **TOL)
    for got, want, n in zip(dev, (a, s, w, sent), "a s w sent".split()):
        torch.testing.assert_close(got.grad.cpu(), want.grad, atol=2e-4 * max(1.0, float(want.grad.abs().max())), rtol=1e-3,
                                   msg=lambda m, n=n: f"d{n}: {m}")


def test_k1_gate_t512_vs_oracle():
    """K1g (attention + sent_linear + sigmoid gate) at T=512, N=25 vs the oracle's un-fused tail (VideoEncoder.py:61-74)."""
    from shufflingvideosfortsg_amd import functional as F
    B = 2
    g = torch.Generator().manual_seed(5)
    mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).requires_grad_(True)
    a, s, w, sent = mk(B, T4, D4), mk(B, N4, D4), mk(D4, sc=D4 ** -0.5), mk(B, N4, D4)
    Wl, bl, r = mk(D4, D4, sc=D4 ** -0.5), mk(D4, sc=0.1), mk(B, T4, D4)
    gO = torch.randn(B, T4, D4, generator=g)
    C0, _ = O.scdm_core(a, s, w, sent)
    out0 = r * torch.sigmoid(torch.nn.functional.linear(C0, Wl, bl))
    out0.backward(gO)
    ref = [x.grad.clone() for x in (a, s, w, sent, Wl, bl, r)]
    dev = [x.detach().cuda().requires_grad_(True) for x in (a, s, w, sent, Wl, bl, r)]
    ad, sd, wd, vd, Wd, bd, rd = dev
    VW = torch.nn.functional.linear(vd, Wd)                       # sent_linear reassociated onto the word rows
    out1 = F.scdm_gate(ad, sd, wd, VW, bd, rd)
    out1.backward(gO.cuda())
    torch.testing.assert_close(out1.detach().cpu(), out0.detach(), **TOL)
    for got, want, n in zip(dev, ref, "a s w sent W_l b_l r".split()):
        torch.testing.assert_close(got.grad.cpu(), want, atol=2e-4 * max(1.0, float(want.abs().max())), rtol=2e-3, msg=lambda m, n=n: f"d{n}: {m}")


@pytest.mark.parametrize("d,heads", [(1024, 8), (2048, 8)])
def test_k2_self_t512_vs_oracle(d, heads):
    """temporal self-attention at T=512: head width 128 (d=1024) and 256 (the Self_Attention_predictor's 2d=2048 over 8 heads)."""
    from shufflingvideosfortsg_amd import functional as F
    B = 1
    g = torch.Generator().manual_seed(d)
    Q, K, V = (torch.randn(B, T4, d, generator=g).requires_grad_(True) for _ in range(3))
    gO = torch.randn(B, T4, d, generator=g)
    o0 = O.mha_core(Q, K, V, heads, d)[0]
    o0.backward(gO)
    dev = [x.detach().cuda().requires_grad_(True) for x in (Q, K, V)]
    o1 = F.mha(*dev, heads, float(d) ** 0.5)
    o1.backward(gO.cuda())
    torch.testing.assert_close(o1.detach().cpu(), o0.detach(), **TOL)
    for got, want, n in zip(dev, (Q, K, V), "QKV"):
        torch.testing.assert_close(got.grad.cpu(), want.grad, atol=2e-4, rtol=2e-3, msg=lambda m, n=n: f"d{n}: {m}")


@pytest.mark.parametrize("mode", ["f32", "f32s"])
def test_lstm_t512_vs_float64(mode, request):
    """one BiLSTM layer at (B=32, T=512, h=512), 512 sequential steps, vs a float64 recurrence."""
    from shufflingvideosfortsg_amd import engine, functional as TF
    engine.set_precision(None if mode == "f32" else "f32s")
    request.addfinalizer(lambda: engine.set_precision(None))
    B, T, h, I = 32, T4, 512, 256
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, T, I, generator=g) * 0.5
    W_ih = torch.randn(8 * h, I, generator=g) / I ** 0.5; W_hh = torch.randn(2, 4 * h, h, generator=g) / h ** 0.5
    bias = torch.randn(8 * h, generator=g) * 0.1
    out, _ = TF.bilstm_layer(x.cuda(), W_ih.cuda(), bias.cuda(), W_hh.cuda(), batch_major=True)
    torch.cuda.synchronize(); TF.check_lstm_errors()
    # float64 reference
    xd, Wd, Hd, bd = x.double(), W_ih.double(), W_hh.double(), bias.double()
    ref = torch.zeros(B, T, 2 * h, dtype=torch.float64)
    for d in range(2):
        hs = torch.zeros(B, h, dtype=torch.float64); cs = torch.zeros_like(hs)
        Gx = xd @ Wd[d * 4 * h:(d + 1) * 4 * h].t() + bd[d * 4 * h:(d + 1) * 4 * h]
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            gt = Gx[:, t] + hs @ Hd[d].t()
            i, f, gg, o = gt.chunk(4, 1)
            cs = torch.sigmoid(f) * cs + torch.sigmoid(i) * torch.tanh(gg)
            hs = torch.sigmoid(o) * torch.tanh(cs)
            ref[:, t, d * h:(d + 1) * h] = hs
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < (2e-5 if mode == "f32s" else 5e-6), err


@pytest.mark.parametrize("dt,bm", [(2, 1), (0, 1), (2, 0)])
def test_lstm_256_rows_run_persistent_in_chunks(dt, bm):
    """B=256 rows at h=512 need 512 workgroups (> 256 CUs): the entry points run two persistent launches over 128-row chunks
    (never the launch-per-step kernels).  Forward outputs / saved state / gate gradients must equal, bit for bit, the two
    128-row halves run on their own, and match the launch-per-step kernels."""
    from shufflingvideosfortsg_amd import _lib, functional as TF
    from shufflingvideosfortsg_amd._lib import ptr
    lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
    TF.check_lstm_errors()
    B, T, h = 256, 24, 512
    g = torch.Generator().manual_seed(77)
    shape = (B, T) if bm else (T, B)
    Gx = (torch.randn(*shape, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
    dOut = torch.randn(*shape, 2 * h, generator=g).cuda(); WT = W.transpose(1, 2).contiguous()

    def run(Gx, dOut, B, persist):
        lib.tsg_lstm_set_persist(persist)
        shape = (B, T) if bm else (T, B)
        sync = torch.zeros(512, dtype=torch.int32, device="cuda")
        out = torch.full((*shape, 2 * h), 9.0, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
        assert lib.tsg_lstm_fwd_bias(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, dt, bm, st) == 0
        nb = lib.tsg_lstm_bwd_ws_bytes(B, T, h)
        assert bool(lib.tsg_lstm_bwd_ws_persistent(B, T, h, nb)) == (persist != 0)
        dG = torch.full((*shape, 2, 4 * h), 5.0, device="cuda"); dC = torch.zeros(2, B, h, device="cuda")
        ws = torch.zeros(nb // 4 + 4, device="cuda"); db = torch.zeros(8 * h, device="cuda")
        assert lib.tsg_lstm_bwd_ws_layout(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, ptr(db), B, T, h, dt, bm, st) == 0
        torch.cuda.synchronize()
        assert int(sync[0]) == 0 and (persist == 0 or int(ws[:1].view(torch.int32)[0]) == 0)
        TF.check_lstm_errors()
        return out, Cs, dG, db
    try:
        full = run(Gx, dOut, B, 1)
        bdim = 0 if bm else 1
        halves = [run(Gx.narrow(bdim, c, 128).contiguous(), dOut.narrow(bdim, c, 128).contiguous(), 128, 1) for c in (0, 128)]
        steps = run(Gx, dOut, B, 0)
    finally:
        lib.tsg_lstm_set_persist(-1)
    for i, name in ((0, "out"), (2, "dG")):
        assert torch.equal(full[i], torch.cat([halves[0][i], halves[1][i]], bdim)), name
    assert torch.equal(full[1], torch.cat([halves[0][1], halves[1][1]], 2)), "Cs"
    torch.testing.assert_close(full[3], halves[0][3] + halves[1][3], atol=1e-4, rtol=1e-5)
    tol = dict(atol=3e-5, rtol=1e-4) if dt == 2 else dict(atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(full[0], steps[0], **tol)
    torch.testing.assert_close(full[2], steps[2], atol=tol["atol"] * 20, rtol=1e-3)


def test_gmd_config4_step_vs_oracle(request):
    """GMD train step at T=512, N=25, d=1024 (B=2 for the CPU oracle), split-precision mode, vs the oracle."""
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    engine.set_precision("f32s")
    request.addfinalizer(lambda: engine.set_precision(None))
    params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, dropout=0.0, video_len=T4, sent_len=N4)
    torch.manual_seed(0)
    model = engine.build_model("gmd", params)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    b = data.synthetic_batch(2, T4, N4, seed=17, pair=True)
    g, pg = b["gt"], b["pseudo_gt"]
    ref = O.gmd_forward(sd, b["query"], b["video"], b["video_mask"], b["pseudo_video"], b["video_mask"],
                        g["temporal_labels"], g["fore_masks"], g["back_masks"], pg["temporal_labels"], pg["fore_masks"], pg["back_masks"])
    ref_loss, _ = O.gmd_losses(ref, b["video_mask"], b["video_mask"], g, pg)
    ref_loss.backward()
    model = model.cuda().train()
    model.tod.dropout.p = 0.0
    d = data.synthetic_batch(2, T4, N4, seed=17, pair=True, device="cuda")
    loss, _, span = engine.gmd_step(model, d, params)
    loss.backward()
    torch.cuda.synchronize(); TF.check_lstm_errors()
    torch.testing.assert_close(span["start"].detach().cpu(), ref[0]["start"].detach(), **TOL)
    torch.testing.assert_close(span["end"].detach().cpu(), ref[0]["end"].detach(), **TOL)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), **TOL)
    for k, p in model.named_parameters():
        want = sd[k].grad
        torch.testing.assert_close(p.grad.cpu(), want, atol=5e-4 * max(1.0, float(want.abs().max())), rtol=5e-3, msg=lambda m, k=k: f"{k}: {m}")


def test_gmd_config4_full_batch_properties(request):
    """B=128 per GPU at T=512 (256 batched encoder rows, the chunked persistent LSTM): one full train step is finite, softmax
    rows sum to one, and duplicated items give bit-identical rows."""
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    engine.set_precision("f32s")
    request.addfinalizer(lambda: engine.set_precision(None))
    B = 128
    params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, dropout=0.0, video_len=T4, sent_len=N4)
    torch.manual_seed(0)
    model = engine.build_model("gmd", params).cuda().train()
    model.tod.dropout.p = 0.0
    d = data.synthetic_batch(B, T4, N4, seed=18, pair=True, device="cuda")
    for k in ("video", "query", "video_mask", "pseudo_video"):
        d[k][B - 1] = d[k][0]                               # item 0 again, in the LAST row (second LSTM chunk)
    for gt in ("gt", "pseudo_gt"):
        for k, v in d[gt].items():
            v[B - 1] = v[0]
    loss, _, span = engine.gmd_step(model, d, params)
    loss.backward()
    torch.cuda.synchronize(); TF.check_lstm_errors()
    assert torch.isfinite(loss)
    for p in (span["start"], span["end"]):
        torch.testing.assert_close(p.sum(1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
        assert torch.equal(p[0], p[B - 1])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
