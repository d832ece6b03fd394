"""BASELINE.json configs[3]: ActivityNet-CD i3d [B=64, T_clip=256, d=1024] train, DP over 4 GPUs (N = 25 words, cfgs/anet_cd_i3d.yml:17-25)
-- the per-GPU shard of 16 pairs.  The GMD train step at T = 256, N = 25 in the f32s mode (the arithmetic bench.py runs) vs the CPU oracle at
B = 2, and one full step at the shard's batch of 16 through the size-independent properties (softmax rows, batch independence, finite fp32
gradients, reproducible loss).  T = 512 (configs[4]) lives in tests/test_config4_gpu.py / test_config5_bf16_gpu.py."""
import pytest
import torch

from oracle import tsg_oracle as O

pytestmark = pytest.mark.gpu
T3, N3 = 256, 25


@pytest.mark.parametrize("mode", [None, "f32s"])
def test_gmd_anet256_step_vs_oracle(mode, request):
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    engine.set_precision(mode)
    request.addfinalizer(lambda: engine.set_precision(None))
    params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, dropout=0.0, video_len=T3, sent_len=N3)
    torch.manual_seed(0)
    model = engine.build_model("gmd", params)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    b = data.synthetic_batch(2, T3, N3, seed=41, pair=True)
    g, pg = b["gt"], b["pseudo_gt"]
    ref = O.gmd_forward(sd, b["query"], b["video"], b["video_mask"], b["pseudo_video"], b["video_mask"],
                        g["temporal_labels"], g["fore_masks"], g["back_masks"], pg["temporal_labels"], pg["fore_masks"], pg["back_masks"])
    ref_loss, _ = O.gmd_losses(ref, b["video_mask"], b["video_mask"], g, pg)
    ref_loss.backward()
    model = model.cuda().train()
    model.tod.dropout.p = 0.0
    d = data.synthetic_batch(2, T3, N3, seed=41, pair=True, device="cuda")
    loss, _, span = engine.gmd_step(model, d, params)
    loss.backward()
    torch.cuda.synchronize(); TF.check_lstm_errors()
    tol = dict(atol=1e-4, rtol=1e-4)                                # the north star's tolerance on boundary scores
    torch.testing.assert_close(span["start"].detach().cpu(), ref[0]["start"].detach(), **tol)
    torch.testing.assert_close(span["end"].detach().cpu(), ref[0]["end"].detach(), **tol)
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), **tol)
    for k, p in model.named_parameters():
        want = sd[k].grad
        torch.testing.assert_close(p.grad.cpu(), want, atol=5e-4 * max(1.0, float(want.abs().max())), rtol=5e-3, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("mode", ["f32s", "bf16"])
def test_gmd_anet256_shard_properties(mode, request):
    from shufflingvideosfortsg_amd import data, engine, functional as TF
    engine.set_precision(mode)
    request.addfinalizer(lambda: engine.set_precision(None))
    B = 16
    params = engine.default_params(video_rnn_hiddendim=512, sent_rnn_hiddendim=512, dropout=0.0, video_len=T3, sent_len=N3)
    torch.manual_seed(0)
    model = engine.build_model("gmd", params).cuda().train()
    model.tod.dropout.p = 0.0
    d = data.synthetic_batch(B, T3, N3, seed=42, pair=True, device="cuda")
    for k in ("video", "query", "video_mask", "pseudo_video"):
        d[k][B - 1] = d[k][0]
    for gt in ("gt", "pseudo_gt"):
        for k, v in d[gt].items():
            v[B - 1] = v[0]
    if mode == "bf16":
        d["video"], d["pseudo_video"] = d["video"].to(torch.bfloat16), d["pseudo_video"].to(torch.bfloat16)
    loss, _, span = engine.gmd_step(model, d, params)
    loss.backward()
    torch.cuda.synchronize(); TF.check_lstm_errors()
    assert torch.isfinite(loss)
    for p in (span["start"], span["end"]):
        torch.testing.assert_close(p.float().sum(1), torch.ones(B, device="cuda"), atol=1e-4, rtol=0)
        assert torch.equal(p[0], p[B - 1]), "batch items are not independent"
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in model.parameters())
    model.zero_grad(set_to_none=True)
    loss2, _, _ = engine.gmd_step(model, d, params)
    torch.cuda.synchronize()
    torch.testing.assert_close(loss2.detach(), loss.detach(), atol=0, rtol=1e-3)
