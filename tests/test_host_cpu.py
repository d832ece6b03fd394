"""Host-side logic on CPU: parameter-name / init-order contract, losses, decode, scorer, data helpers."""
import json
import logging
import os

import numpy as np
import pytest
import torch

from shufflingvideosfortsg_amd import IoU_eval, data, engine
from shufflingvideosfortsg_amd import loss as L
from shufflingvideosfortsg_amd.model.networks import attention as A

TOL = dict(atol=2e-6, rtol=1e-5)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_contract():
    """Checkpoint compatibility: same keys, order and shapes as the reference GMD."""
    with open(os.path.join(GOLD, "gmd_state_dict_contract.json")) as f:
        want = json.load(f)
    sd = engine.build_model("gmd", engine.default_params()).state_dict()
    assert list(sd.keys()) == list(want.keys())
    assert all(list(sd[k].shape) == want[k] for k in want)
    base = engine.build_model("qave", engine.default_params()).state_dict()
    assert list(base.keys()) == [k for k in want if not k.startswith(("csmm.", "tod."))]


def test_default_init_matches_reference(golden):
    """Same seed -> same weights as the reference model (construction order + PyTorch default init):
    per-tensor checksums captured from the real reference at BASELINE config 0 (d=512)."""
    g = golden("config0")
    torch.manual_seed(0)
    model = engine.build_model("qave", engine.default_params())
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g.a["keys"]]
    np.testing.assert_allclose([float(sd[k].double().sum()) for k in sd], g.a["wsum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([float(sd[k].double().abs().sum()) for k in sd], g.a["wabs"], rtol=0, atol=1e-9)


def test_losses_and_decode(golden):
    g = golden("losses")
    s, e = g.t("start"), g.t("end")
    torch.testing.assert_close(L.span_ground_loss(s, e, g.a["fs"]), g.t("span_ground"), **TOL)
    torch.testing.assert_close(L.BCE_loss(g.t("logits"), g.t("labels"), g.t("mask")), g.t("bce"), **TOL)
    kl = L.matching_KL_divergence(A.masked_softmax(g.t("logits"), g.t("labels")),
                                  A.masked_softmax(g.t("logits2"), g.t("labels2")), g.a["fs"], g.a["fs2"])
    torch.testing.assert_close(kl, g.t("kl"), **TOL)
    torch.testing.assert_close(L.temporal_order_discrimination_loss(g.t("od"), g.t("pd")), g.t("tod"), **TOL)
    pred, score = L.span_pred(s, e)
    assert torch.equal(pred, g.t("pred"))
    torch.testing.assert_close(score, g.t("score"), **TOL)
    torch.testing.assert_close(L.compute_mean_iou(pred.float(), g.t("seg2")), g.t("miou"), **TOL)


def test_span_pred_ties_and_edges():
    # first maximum wins; a single clip; end before start is never chosen (upper triangle only)
    p = torch.tensor([[0.25, 0.25, 0.25, 0.25]])
    assert L.span_pred(p, p)[0].tolist() == [[0, 0]]
    assert L.span_pred(torch.ones(2, 1), torch.ones(2, 1))[0].tolist() == [[0, 0], [0, 0]]
    s = torch.tensor([[0.0, 0.0, 1.0]]); e = torch.tensor([[0.9, 0.0, 0.1]])
    assert L.span_pred(s, e)[0].tolist() == [[2, 2]]          # (2,0) would score 1.9 but is below the diagonal


def test_mask_helpers_and_posenc(golden):
    g = golden("mask_helpers")
    torch.testing.assert_close(A.masked_softmax(g.t("vec"), g.t("mask")), g.t("masked_softmax"), **TOL)
    torch.testing.assert_close(A.mask_logits(g.t("vec"), g.t("mask")), g.t("mask_logits"), atol=0, rtol=0)
    torch.testing.assert_close(A.mask_logits(g.t("feat"), g.t("mask"), 0.0), g.t("mask_logits3"), atol=0, rtol=0)
    g = golden("posenc")
    x = torch.zeros(1, int(g.a["T"]), int(g.a["D"]))
    torch.testing.assert_close(A.positional_encodings_like(x), g.t("enc"), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("name", ["charades_cd", "anet_cd"])
def test_iou_known_answers(golden, name):
    """The reference's committed prediction files re-scored = the numbers in its test logs."""
    g = golden("iou")
    miou, recall = IoU_eval.score(g.a[name + "_pred"], g.a[name + "_gt"])
    np.testing.assert_allclose([miou] + recall, g.a[name + "_logged"], atol=1e-9)
    # through the submits-JSON schema
    sub = {"version": "V0", "external_data": {}, "results": {
        f"v{i}": [{"timestamp": p.tolist(), "gt_timestamp": q.tolist()}]
        for i, (p, q) in enumerate(zip(g.a[name + "_pred"][:200], g.a[name + "_gt"][:200]))}}
    assert IoU_eval.retrieval_eval(sub, verbose=False) == IoU_eval.score(g.a[name + "_pred"][:200], g.a[name + "_gt"][:200])


def test_aug_and_masks(golden):
    g = golden("aug")
    for i in range(len(g.a["fs"])):
        nf, n, nv = data.gt_moment_translate(g.a["fs"][i].tolist(), int(g.a["nfeats"][i]), g.a["video"][i], int(g.a["pos"][i]))
        assert list(nf) == g.a["new_fs"][i].tolist()
        np.testing.assert_array_equal(nv, g.a["new_video"][i])
    for b, m in zip(g.a["seqmask_b"], g.a["seqmask"]):
        np.testing.assert_array_equal(data.Sequence_mask(10, b.tolist()), m)
    b = data.synthetic_batch(4, 16, 5, video_dim=8, pair=True)
    for (s, e), (ps, pe) in zip(b["gt"]["framestps"], b["pseudo_gt"]["framestps"]):
        assert e - s == pe - ps                                  # the moment keeps its length
    assert b["pseudo_video"].shape == b["video"].shape


def test_unsupported_predictor_is_loud():
    p = engine.default_params(predictor="tied_lstm")
    with pytest.raises(NotImplementedError):
        engine.build_model("qave", p, logging.getLogger("t"))


def test_strong_scaling_shards_and_device_move():
    """bench.py --scaling strong: every rank builds the same global batch and keeps its contiguous shard (dp.shard_batch, the
    reference's DataParallel scatter, train.py:343); data.to_device turns the per-sample [start, end] lists into one index tensor."""
    import torch
    from shufflingvideosfortsg_amd import data
    from shufflingvideosfortsg_amd.dp import shard_batch
    full = data.synthetic_batch(8, 16, 5, seed=3, pair=True)
    parts = [shard_batch(full, r, 4) for r in range(4)]
    for k in ("video", "query", "video_mask", "pseudo_video"):
        assert torch.equal(torch.cat([p[k] for p in parts]), full[k])
    for gt in ("gt", "pseudo_gt"):
        assert sum((p[gt]["framestps"] for p in parts), []) == full[gt]["framestps"]
        assert torch.equal(torch.cat([p[gt]["temporal_labels"] for p in parts]), full[gt]["temporal_labels"])
    d = data.to_device(parts[1], "cpu")
    assert d["gt"]["framestps"].dtype == torch.long and d["gt"]["framestps"].tolist() == full["gt"]["framestps"][2:4]
    assert isinstance(parts[1]["gt"]["framestps"], list)                      # the host batch is not modified


def test_pair_batch_halves_are_adjacent_and_cat_is_a_view():
    """data.to_device places the original and the shuffled stream back to back; adjacent_cat then returns a view with torch.cat's
    values, and falls back to a copy for anything else (other storages, a gap, a tensor that needs a gradient)."""
    import torch
    from shufflingvideosfortsg_amd import data
    b = data.synthetic_batch(4, 16, 5, video_dim=32, pair=True)
    d = data.to_device(b, "cpu")
    v = data.adjacent_cat(d["video"], d["pseudo_video"])
    assert v.data_ptr() == d["video"].data_ptr() and torch.equal(v, torch.cat([b["video"], b["pseudo_video"]], 0))
    for k in ("temporal_labels", "fore_masks", "back_masks"):
        m = data.adjacent_cat(d["gt"][k], d["pseudo_gt"][k])
        assert m.data_ptr() == d["gt"][k].data_ptr() and torch.equal(m, torch.cat([b["gt"][k], b["pseudo_gt"][k]], 0))
    x, y = torch.randn(3, 4), torch.randn(3, 4)
    assert torch.equal(data.adjacent_cat(x, y), torch.cat([x, y], 0))
    z = torch.randn(6, 4)
    c = data.adjacent_cat(z[:2], z[3:5])                                    # same storage, a gap between them: a copy
    assert c.data_ptr() != z.data_ptr() and torch.equal(c, torch.cat([z[:2], z[3:5]], 0))
    g = torch.randn(4, 4, requires_grad=True)
    c = data.adjacent_cat(g[:2], g[2:])                                     # needs a gradient: torch.cat (autograd splits it back)
    c.sum().backward()
    assert torch.equal(g.grad, torch.ones(4, 4))


def test_runtime_env_is_set_before_hip_starts():
    """Importing the package sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (the ROCm runtime's graph packet capture lets replayed graphs read stale data from
    reused pool blocks: _runtime_env.py); an explicit other value is respected and reported as unsafe, which GraphedTrainStep refuses."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import shufflingvideosfortsg_amd as P; "
            "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), P._runtime_env.graph_replay_safe())" % root)
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.split() == ["0", "True"]
    assert subprocess.run([sys.executable, "-c", code], env=dict(env, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1"), capture_output=True, text=True).stdout.split() == ["1", "False"]
