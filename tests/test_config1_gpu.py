"""BASELINE config 1 -- synthetic [B=32, T_clip=64, T_word=20, d=512], forward only, strict fp32 on one MI355X -- at the
level of every kernel the forward launches (K1g, K3, the persistent BiLSTM at h=256) and of the assembled QAVE / GMD
forward (VERDICT r2 "missing" #3: only plain K1 at (4,64,20,512) ran before).

The CPU oracle cannot run 32 pairs in seconds, so every test follows the full-size pattern: the launch is made at the FULL
config-1 batch (the grid, tile and row-per-workgroup choices of the kernels depend on B), a few items are checked against the
oracle run on those items only, and size-independent properties are checked over all 32 pairs."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)          # north-star: boundary scores within 1e-4 fp32
B1, T1, N1, D1 = 32, 64, 20, 512          # BASELINE.json configs[1]
NREF = 3                                  # items the oracle re-computes


def test_k1_gate_config1_vs_oracle():
    """K1g (tsg_scdm_gate_fwd) at [32,64,20,512]: items 0..2 vs the oracle's un-fused recalibration tail; softmax rows; duplicated
    item bit-identical; a launch of the 3 items alone (another grid / rows-per-workgroup choice) agrees to rounding."""
    from shufflingvideosfortsg_amd import functional as F
    lin = torch.nn.functional.linear
    B, T, N, d = B1, T1, N1, D1
    g = torch.Generator().manual_seed(101)
    r = torch.randn(B, T, d, generator=g); word = torch.randn(B, N, d, generator=g)
    r[B - 1] = r[0]; word[B - 1] = word[0]
    p = {k: torch.randn(*sh, generator=g) / d ** 0.5 for k, sh in dict(Ws=(d, d), Wa=(d, d), ba=(d,), w=(1, d), Wl=(d, d), bl=(d,)).items()}
    C = O.scdm_attention(r[:NREF], word[:NREF], p["Ws"], p["Wa"], p["ba"], p["w"])
    want = r[:NREF] * torch.sigmoid(lin(C, p["Wl"], p["bl"]))
    rd, wd = r.cuda(), word.cuda()
    pd = {k: v.cuda() for k, v in p.items()}
    a, s, VW = lin(rd, pd["Wa"], pd["ba"]), lin(wd, pd["Ws"]), lin(wd, pd["Wl"])
    out = F.scdm_gate(a, s, pd["w"], VW, pd["bl"], rd)
    C1, P1 = F.scdm_attn(a, s, pd["w"], wd, return_p=True)
    torch.cuda.synchronize()
    torch.testing.assert_close(out[:NREF].cpu(), want, **TOL)
    torch.testing.assert_close(C1[:NREF].cpu(), C, **TOL)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(P1.sum(-1), torch.ones(B, T, device="cuda"), atol=1e-5, rtol=0)
    assert torch.equal(out[0], out[B - 1]) and torch.equal(P1[0], P1[B - 1])
    small = F.scdm_gate(a[:NREF].contiguous(), s[:NREF].contiguous(), pd["w"], VW[:NREF].contiguous(), pd["bl"], rd[:NREF].contiguous())
    torch.testing.assert_close(small, out[:NREF], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("use_mask", [False, True])
def test_k3_config1_vs_oracle(use_mask):
    """K3 (tsg_boundary_score_fwd) at [32,64, Dv=Ds=512, Hm=256]: items 0..2 vs the oracle's concat + MLP_predictor; rows sum to 1;
    masked clips get exactly 0."""
    from shufflingvideosfortsg_amd import functional as F
    from test_boundary_gpu import _stack, _weights
    B, T, d, Hm = B1, T1, D1, 256
    g = torch.Generator().manual_seed(102)
    p = _weights(d, d, Hm, g)
    video = torch.randn(B, T, d, generator=g); sent = torch.randn(B, d, generator=g)
    video[B - 1] = video[0]; sent[B - 1] = sent[0]
    mask = None
    if use_mask:
        n = torch.randint(T // 2, T + 1, (B,), generator=g)
        n[B - 1] = n[0]
        mask = (torch.arange(T)[None, :] < n[:, None]).int()
    s0, e0 = O.mlp_predictor(O.video_sentence_concat(video[:NREF], sent[:NREF]), p, mask[:NREF] if use_mask else None)
    dev = {k: v.cuda() for k, v in p.items()}
    W1v, W1s, b1, w2, b2 = _stack(dev, d)
    y = torch.nn.functional.linear(video.cuda(), W1v)
    cs = torch.nn.functional.linear(sent.cuda(), W1s)
    s1, e1 = F.boundary_score(y, cs, b1, w2, b2, None, mask.cuda() if use_mask else None)
    torch.cuda.synchronize()
    torch.testing.assert_close(s1[:NREF].cpu(), s0, **TOL)
    torch.testing.assert_close(e1[:NREF].cpu(), e0, **TOL)
    for q in (s1, e1):
        assert torch.isfinite(q).all()
        torch.testing.assert_close(q.sum(1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
        assert torch.equal(q[0], q[B - 1])
        if use_mask:
            assert (q[mask.cuda() == 0] == 0).all()


def test_bilstm_config1_vs_oracle():
    """The persistent BiLSTM forward at the config-1 video shape (B=32 rows, T=64, 1024 -> h=256, 2 layers), strict fp32:
    items 0..2 vs the oracle's explicit recurrence, duplicated row bit-identical."""
    from shufflingvideosfortsg_amd import functional as TF
    from shufflingvideosfortsg_amd.model.networks.RNN import BiLSTM
    from test_lstm_gpu import _params
    B, T, I, h = B1, T1, 1024, D1 // 2
    g = torch.Generator().manual_seed(103)
    p = _params(I, h, 2, g)
    x = torch.randn(B, T, I, generator=g)
    x[B - 1] = x[0]
    out0, hn0, cn0 = O.bilstm(x[:NREF], p, 2)
    m = BiLSTM(I, h, 2, 0.0)
    m.load_state_dict(p)
    m.cuda().eval()
    assert m.backend == "hip"
    with torch.no_grad():
        out1, hn1, cn1 = m(x.cuda())
    torch.cuda.synchronize(); TF.check_lstm_errors()
    torch.testing.assert_close(out1[:NREF].cpu(), out0, **TOL)
    torch.testing.assert_close(hn1[:, :NREF].cpu(), hn0, **TOL)
    torch.testing.assert_close(cn1[:, :NREF].cpu(), cn0, **TOL)
    assert torch.isfinite(out1).all() and torch.equal(out1[0], out1[B - 1])


@pytest.mark.parametrize("kind", ["qave", "gmd"])
def test_model_forward_config1_vs_oracle(kind):
    """The assembled forward at config 1 (default init under manual_seed(0), d=512 = the reference's default widths), forward only,
    strict fp32: QAVE `Baseline.forward` and `GMD.eval_forward` (the test-time path, SpanGroundMatchDisc.py:101-129) on all 32 pairs;
    boundary scores and decoded spans of items 0..2 vs the oracle; properties over the batch."""
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    from shufflingvideosfortsg_amd import loss as L
    B, T, N = B1, T1, N1
    params = engine.default_params(video_rnn_hiddendim=D1 // 2, sent_rnn_hiddendim=D1 // 2, dropout=0.0, video_len=T, sent_len=N)
    torch.manual_seed(0)
    model = engine.build_model(kind, params)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cpu = data.synthetic_batch(B, T, N, seed=1234, pair=False)
    for k in ("video", "query", "video_mask", "query_mask"):
        cpu[k][B - 1] = cpu[k][0]
    with torch.no_grad():
        if kind == "gmd":
            ref = O.gmd_eval_forward(sd, cpu["video"][:NREF], cpu["query"][:NREF], cpu["video_mask"][:NREF])
        else:
            ref = O.baseline_forward(sd, cpu["video"][:NREF], cpu["query"][:NREF], cpu["video_mask"][:NREF])
    model = model.cuda().eval()
    engine.set_precision(None)
    fwd = model.eval_forward if kind == "gmd" else model
    with torch.no_grad():
        out = fwd(cpu["video"].cuda(), cpu["query"].cuda(), cpu["video_mask"].cuda(), cpu["query_mask"].cuda())
    torch.cuda.synchronize(); TF.check_lstm_errors()
    for k in ("start", "end"):
        torch.testing.assert_close(out[k][:NREF].cpu(), ref[k], **TOL, msg=lambda m, k=k: f"{k}: {m}")
        assert torch.isfinite(out[k]).all()
        torch.testing.assert_close(out[k].sum(1), torch.ones(B, device="cuda"), atol=1e-5, rtol=0)
        assert torch.equal(out[k][0], out[k][B - 1]), "batch items are not independent"
    # decode on the device (tsg_span_pred) == the oracle's span_pred on the SAME probabilities, bit for bit, all 32 pairs; the
    # best score of the oracle's own probabilities agrees to the score tolerance (at default init the rows are nearly flat, so the
    # argmax of two results 1e-7 apart may legitimately differ: the indices are compared on identical inputs only)
    pred, score = L.span_pred(out["start"], out["end"])
    pref, sref = O.span_pred(out["start"].cpu(), out["end"].cpu())
    assert torch.equal(pred.cpu().long(), pref.long()) and torch.equal(score.cpu(), sref)
    torch.testing.assert_close(score[:NREF].cpu(), O.span_pred(ref["start"], ref["end"])[1], **TOL)
