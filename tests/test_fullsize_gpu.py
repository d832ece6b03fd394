"""The full-size GMD train step of bench.py -- [B=64, T_clip=128, T_word=20, d=1024], default split-precision mode --
checked against the CPU oracle and through size-independent properties, plus the K5 matching head
against the oracle's csmm."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)          # north-star: boundary scores within 1e-4 fp32


def _sub(batch, sl, device=None):
    """rows `sl` of a (possibly device) batch dict as CPU tensors (or on ``device``)"""
    def cut(v):
        if isinstance(v, torch.Tensor):
            v = v[sl]
            return v.cpu() if device is None else v.to(device)
        return v[sl]
    out = {}
    for k, v in batch.items():
        out[k] = {kk: cut(vv) for kk, vv in v.items()} if isinstance(v, dict) else cut(v)
    return out


BF16_TOL = dict(atol=1e-2, rtol=2e-2)     # the bf16 GEMM mode has its own tolerance (~1e-2 relative)


@pytest.mark.parametrize("gemm", ["f32s", None, "bf16", "bf16-storage"])
def test_full_size_gmd_step_vs_oracle(gemm, request):
    """engine.gmd_step at the bench shape; the oracle cannot run 64 pairs in seconds, so:
      * boundary scores / matching logits / discriminator logits of 2 batch items vs the oracle on those items (1e-4);
      * the losses restricted to those items, formed from the FULL-SIZE forward's outputs, are back-propagated through the
        full-size backward kernels (all other items receive zero upstream gradient): every parameter gradient must equal the
        oracle's 2-item gradients;
      * properties over all 64 pairs: softmax rows of start / end sum to 1, finite outputs, batch independence (items 2 and 3
        are duplicates of 0 and 1 -> bit-identical rows), and the full 64-pair loss / backward are finite."""
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    from shufflingvideosfortsg_amd import loss as L
    from shufflingvideosfortsg_amd.model.networks.attention import masked_softmax
    # "bf16": library GEMM operands bf16 (autocast), fp32 accumulate, fp32 storage; "bf16-storage": the TSG_BF16 path -- bf16
    # activations in HBM through every hand-written kernel (BASELINE configs 2 / 4), fp32 arithmetic inside, fp32 master weights
    storage = gemm == "bf16-storage"
    bf16 = gemm == "bf16" or storage
    mode = "bf16" if storage else (torch.bfloat16 if bf16 else gemm)
    tol = BF16_TOL if bf16 else TOL
    precision = lambda: engine.precision(mode)
    request.addfinalizer(lambda: engine.set_precision(None))
    B, T, N = 64, 128, 20
    params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, dropout=0.0, video_len=T, sent_len=N)
    torch.manual_seed(0)
    model = engine.build_model("gmd", params)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    cpu = data.synthetic_batch(B, T, N, seed=99, pair=True)
    # duplicate items 0,1 into 2,3 (batch independence)
    for k in ("video", "query", "video_mask", "query_mask", "pseudo_video"):
        cpu[k][2:4] = cpu[k][0:2]
    for gt in ("gt", "pseudo_gt"):
        for k, v in cpu[gt].items():
            if isinstance(v, torch.Tensor):
                v[2:4] = v[0:2]
            else:
                v[2:4] = [list(x) for x in v[0:2]]
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in cpu.items() if not isinstance(v, dict)}
    if storage:                           # the clip features are resident in the storage dtype (as bench.py --dtype bf16 holds them)
        dev["video"], dev["pseudo_video"] = dev["video"].to(torch.bfloat16), dev["pseudo_video"].to(torch.bfloat16)
    for gt in ("gt", "pseudo_gt"):
        dev[gt] = {k: (v.cuda() if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=torch.long).cuda()) for k, v in cpu[gt].items()}

    # oracle on items 0..1
    s2 = _sub(cpu, slice(0, 2))
    g, pg = s2["gt"], s2["pseudo_gt"]
    ref = O.gmd_forward(sd, s2["query"], s2["video"], s2["video_mask"], s2["pseudo_video"], s2["video_mask"],
                        g["temporal_labels"], g["fore_masks"], g["back_masks"], pg["temporal_labels"], pg["fore_masks"], pg["back_masks"])
    ref_loss, _ = O.gmd_losses(ref, s2["video_mask"], s2["video_mask"], g, pg)
    ref_loss.backward()

    model = model.cuda().train()
    model.tod.dropout.p = 0.0
    dg, dpg = dev["gt"], dev["pseudo_gt"]
    with precision():
        span, om, pm, od, pd = model(dev["query"], dev["query_mask"], dev["video"], dev["video_mask"], dev["pseudo_video"], dev["video_mask"],
                                     dg["temporal_labels"], dg["fore_masks"], dg["back_masks"],
                                     dpg["temporal_labels"], dpg["fore_masks"], dpg["back_masks"])
    span = {k: v.float() for k, v in span.items()}
    om, pm, od, pd = om.float(), pm.float(), od.float(), pd.float()
    torch.cuda.synchronize()
    TF.check_lstm_errors()
    # (1) two items vs the oracle
    for got, want, name in ((span["start"], ref[0]["start"], "start"), (span["end"], ref[0]["end"], "end"), (om, ref[1], "ori_match"),
                            (pm, ref[2], "pseudo_match"), (od, ref[3], "ori_disc"), (pd, ref[4], "pseudo_disc")):
        torch.testing.assert_close(got[:2].detach().cpu(), want.detach(), **tol, msg=lambda m, n=name: f"{n}: {m}")
    # (3) properties over the whole batch
    for p in (span["start"], span["end"]):
        assert torch.isfinite(p).all()
        torch.testing.assert_close(p.sum(1), torch.ones(B, device="cuda"), atol=1e-3 if bf16 else 1e-5, rtol=0)
    for t in (span["start"], span["end"], om, pm, od, pd):
        assert torch.equal(t[0:2], t[2:4]), "batch items are not independent"
    # (2) the 2-item losses through the full-size backward
    vm = dev["video_mask"][:2]
    fs, pfs = dg["framestps"][:2], dpg["framestps"][:2]
    lsub = (L.span_ground_loss(span["start"][:2], span["end"][:2], fs)
            + L.BCE_loss(om[:2], dg["temporal_labels"][:2], vm) + L.BCE_loss(pm[:2], dpg["temporal_labels"][:2], vm)
            + L.matching_KL_divergence(masked_softmax(om[:2], dg["temporal_labels"][:2]), masked_softmax(pm[:2], dpg["temporal_labels"][:2]), fs, pfs)
            + L.temporal_order_discrimination_loss(od[:2], pd[:2]))
    torch.testing.assert_close(lsub.detach().cpu(), ref_loss.detach(), **tol)
    lsub.backward()
    torch.cuda.synchronize()
    TF.check_lstm_errors()
    for k, p in model.named_parameters():
        want = sd[k].grad
        atol = (3e-2 if bf16 else 5e-4) * max(1.0, float(want.abs().max()))
        torch.testing.assert_close(p.grad.cpu(), want, atol=atol, rtol=1e-1 if bf16 else 5e-3, msg=lambda m, k=k: f"grad {k}: {m}")
    # the fused full-batch step (K4 losses) is finite and its backward too
    model.zero_grad(set_to_none=True)
    with precision():
        loss, parts, _ = engine.gmd_step(model, dev, params)
    loss.backward()
    torch.cuda.synchronize()
    TF.check_lstm_errors()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("B,T,d,Hm", [(3, 40, 64, 128), (2, 128, 1024, 1024)])
def test_match_head_module_vs_oracle(B, T, d, Hm):
    """csmm = VideoTextSemanticMatch (split-W first Linear + K5 tail) vs the oracle's csmm (DistributionAlign.py:97-118), outputs
    and every gradient."""
    from shufflingvideosfortsg_amd.model.components.DistributionAlign import VideoTextSemanticMatch
    torch.manual_seed(3)
    m = VideoTextSemanticMatch(dict(name="concat", video_dim=d, query_dim=d), dict(name="none", hidden_dim=256, layers=2, dropout=0.0),
                               dict(name="mlp", activation="relu", hidden_dim=Hm))
    g = torch.Generator().manual_seed(B * T)
    video = torch.randn(B, T, d, generator=g); sent = torch.randn(B, d, generator=g); gl = torch.randn(B, T, generator=g)
    w = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    v0, s0 = video.clone().requires_grad_(True), sent.clone().requires_grad_(True)
    ref = O.csmm(v0, s0, w)
    ref.backward(gl)
    m = m.cuda()
    v1, s1 = video.cuda().requires_grad_(True), sent.cuda().requires_grad_(True)
    out, _ = m(v1, s1, None)
    out.backward(gl.cuda())
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), **TOL)
    torch.testing.assert_close(v1.grad.cpu(), v0.grad, atol=2e-4, rtol=2e-3)
    torch.testing.assert_close(s1.grad.cpu(), s0.grad, atol=2e-4 * max(1.0, float(s0.grad.abs().max())), rtol=2e-3)
    for k, p in m.named_parameters():
        want = w[k].grad
        torch.testing.assert_close(p.grad.cpu(), want, atol=2e-4 * max(1.0, float(want.abs().max())), rtol=2e-3, msg=lambda s, k=k: f"{k}: {s}")
