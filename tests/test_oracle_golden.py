"""Pin oracle/tsg_oracle.py against golden vectors captured from the real reference
(oracle/make_golden.py).  fp32, CPU; tolerance 2e-6 abs unless stated (different but equivalent
op orders: vectorised vs the reference's per-word / per-head Python loops)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import tsg_oracle as O

TOL = dict(atol=2e-6, rtol=1e-5)


def leafs(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def check_grads(sd, want, **tol):
    tol = tol or TOL
    for k, g in want.items():
        assert sd[k].grad is not None, k
        torch.testing.assert_close(sd[k].grad, g, **tol, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_scdm(golden, tag):
    g = golden("scdm_" + tag)
    w = leafs(g.weights)
    v = g.t("video").requires_grad_(True); s = g.t("sent").requires_grad_(True)
    C = O.scdm_attention(v, s, w["W_s.weight"], w["W_a.weight"], w["W_a.bias"], w["w.weight"])
    torch.testing.assert_close(C, g.t("C"), **TOL)
    C.backward(g.t("gC"))
    torch.testing.assert_close(v.grad, g.t("gvideo"), **TOL)
    torch.testing.assert_close(s.grad, g.t("gsent"), **TOL)
    check_grads(w, g.wgrads)
    # the fused-kernel core on projected inputs is the same function
    a = torch.nn.functional.linear(v, w["W_a.weight"], w["W_a.bias"]); sp = torch.nn.functional.linear(s, w["W_s.weight"])
    C2, P = O.scdm_core(a, sp, w["w.weight"].reshape(-1), s)
    torch.testing.assert_close(C2, g.t("C"), **TOL)
    torch.testing.assert_close(P.sum(-1), torch.ones_like(P.sum(-1)), **TOL)


@pytest.mark.parametrize("tag", ["cross", "self", "causal", "onehead"])
def test_multihead(golden, tag):
    g = golden("mha_" + tag)
    w = leafs(g.weights)
    h, causal = int(g.a["n_heads"]), bool(g.a["causal"])
    q = g.t("q").requires_grad_(True)
    if tag in ("self", "causal"):
        k = v = q
    else:
        k = g.t("k").requires_grad_(True); v = g.t("v").requires_grad_(True)
    out, A, S = O.multihead(q, k, v, w["wq.weight"], w["wk.weight"], w["wv.weight"], w["wo.weight"], h, causal, True)
    torch.testing.assert_close(out, g.t("out"), **TOL)
    # causal A holds -1e10/sqrt(d) entries: compare with a relative tolerance
    torch.testing.assert_close(A, g.t("A"), atol=2e-5, rtol=1e-6)
    torch.testing.assert_close(S, g.t("A_softmax"), **TOL)
    out.backward(g.t("g"))
    torch.testing.assert_close(q.grad, g.t("gq"), **TOL)
    if tag not in ("self", "causal"):
        torch.testing.assert_close(k.grad, g.t("gk"), **TOL)
        torch.testing.assert_close(v.grad, g.t("gv"), **TOL)
    check_grads(w, g.wgrads)
    # F2: the scale is sqrt(d_model), i.e. SDPA with scale 1/sqrt(d_model) per head
    d = q.shape[-1]
    Q = torch.nn.functional.linear(q, w["wq.weight"]); K = torch.nn.functional.linear(k, w["wk.weight"])
    V = torch.nn.functional.linear(v, w["wv.weight"])
    O2, A2, S2 = O.mha_core(Q, K, V, h, d, causal)
    torch.testing.assert_close(torch.nn.functional.linear(O2, w["wo.weight"]), g.t("out"), **TOL)


def test_attention_causal(golden):
    g = golden("attention_causal")
    o, A, S = O.attention(g.t("q"), g.t("k"), g.t("v"), 16, True)
    torch.testing.assert_close(o, g.t("out"), **TOL)
    torch.testing.assert_close(A, g.t("A"), atol=1e-5, rtol=1e-6)
    torch.testing.assert_close(S, g.t("S"), **TOL)


def test_posenc_and_masks(golden):
    g = golden("posenc")
    torch.testing.assert_close(O.positional_encodings(int(g.a["T"]), int(g.a["D"])), g.t("enc"), atol=0, rtol=0)
    g = golden("mask_helpers")
    torch.testing.assert_close(O.masked_softmax(g.t("vec"), g.t("mask")), g.t("masked_softmax"), **TOL)
    torch.testing.assert_close(O.mask_logits(g.t("vec"), g.t("mask")), g.t("mask_logits"), atol=0, rtol=0)
    torch.testing.assert_close(O.mask_logits(g.t("vec"), g.t("mask"), 0.0), g.t("mask_logits0"), atol=0, rtol=0)
    torch.testing.assert_close(O.mask_logits(g.t("feat"), g.t("mask"), 0.0), g.t("mask_logits3"), atol=0, rtol=0)


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_mlp_predictor(golden, tag):
    g = golden("mlp_" + tag)
    w = leafs(g.weights)
    x = g.t("x").requires_grad_(True)
    s, e = O.mlp_predictor(x, w, g.t("mask") if tag == "mask" else None)
    torch.testing.assert_close(s, g.t("start"), **TOL)
    torch.testing.assert_close(e, g.t("end"), **TOL)
    (s * g.t("gs") + e * g.t("ge")).sum().backward()
    torch.testing.assert_close(x.grad, g.t("gx"), **TOL)
    check_grads(w, g.wgrads)


def test_concat(golden):
    g = golden("concat")
    torch.testing.assert_close(O.video_sentence_concat(g.t("video"), g.t("sent")), g.t("cross"), atol=0, rtol=0)


@pytest.mark.parametrize("tag", ["pe", "nope"])
def test_self_attention_predictor(golden, tag):
    g = golden("selfattn_pred_" + tag)
    w = leafs(g.weights)
    x = g.t("x").requires_grad_(True)
    s, e = O.self_attention_predictor(x, w, int(g.a["n_heads"]), tag == "pe")
    torch.testing.assert_close(s, g.t("start"), **TOL)
    torch.testing.assert_close(e, g.t("end"), **TOL)
    (s * g.t("gs") + e * g.t("ge")).sum().backward()
    torch.testing.assert_close(x.grad, g.t("gx"), **TOL)
    check_grads({k: v for k, v in w.items()}, g.wgrads)


def test_bilstm(golden):
    g = golden("bilstm")
    w = leafs(g.weights)
    x = g.t("x").requires_grad_(True)
    out, hn, cn = O.bilstm(x, w, 2)
    torch.testing.assert_close(out, g.t("out"), **TOL)
    torch.testing.assert_close(hn, g.t("hn"), **TOL)
    torch.testing.assert_close(cn, g.t("cn"), **TOL)
    ((out * g.t("g")).sum() + (hn * g.t("gh")).sum()).backward()
    torch.testing.assert_close(x.grad, g.t("gx"), **TOL)
    check_grads(w, g.wgrads, atol=5e-6, rtol=1e-5)


def test_query_aware_encoder(golden):
    g = golden("qave")
    w = leafs(g.weights)
    v = g.t("video").requires_grad_(True); word = g.t("word").requires_grad_(True)
    o = O.query_aware_encoder(v, word, w)
    torch.testing.assert_close(o, g.t("out"), atol=5e-6, rtol=1e-5)
    o.backward(g.t("g"))
    torch.testing.assert_close(v.grad, g.t("gvideo"), atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(word.grad, g.t("gword"), atol=5e-6, rtol=1e-5)
    check_grads(w, g.wgrads, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_baseline(golden, tag):
    g = golden("baseline_" + tag)
    w = leafs(g.weights)
    out = O.baseline_forward(w, g.t("video"), g.t("query"), g.t("vmask"), use_mask=(tag == "mask"))
    torch.testing.assert_close(out["start"], g.t("start"), **TOL)
    torch.testing.assert_close(out["end"], g.t("end"), **TOL)
    loss = O.span_ground_loss(out["start"], out["end"], g.a["framestps"])
    torch.testing.assert_close(loss, g.t("loss"), **TOL)
    loss.backward()
    check_grads(w, g.wgrads, atol=1e-5, rtol=1e-4)
    pred, score = O.span_pred(out["start"].detach(), out["end"].detach())
    assert torch.equal(pred, g.t("pred"))
    torch.testing.assert_close(score, g.t("score"), **TOL)


def test_gmd(golden):
    g = golden("gmd")
    w = leafs(g.weights)
    out = O.gmd_forward(w, g.t("query"), g.t("video"), g.t("vmask"), g.t("pvideo"), g.t("vmask"),
                        g.t("ot"), g.t("of"), g.t("ob"), g.t("pt"), g.t("pf"), g.t("pb"))
    span, om, pm, od, pd = out
    for got, key in ((span["start"], "start"), (span["end"], "end"), (om, "om"), (pm, "pm"), (od, "od"), (pd, "pd")):
        torch.testing.assert_close(got, g.t(key), atol=5e-6, rtol=1e-5, msg=lambda m, key=key: f"{key}: {m}")
    ogt = {"framestps": g.a["framestps"].tolist(), "temporal_labels": g.t("ot")}
    pgt = {"framestps": g.a["pframestps"].tolist(), "temporal_labels": g.t("pt")}
    loss, (lg, l1, l2, ld) = O.gmd_losses(out, g.t("vmask"), g.t("vmask"), ogt, pgt)
    for got, key in ((lg, "lg"), (l1, "l1"), (l2, "l2"), (ld, "ld"), (loss, "loss")):
        torch.testing.assert_close(got, g.t(key), atol=5e-6, rtol=1e-5, msg=lambda m, key=key: f"{key}: {m}")
    loss.backward()
    check_grads(w, g.wgrads, atol=2e-5, rtol=1e-4)
    ev = O.gmd_eval_forward(w, g.t("video"), g.t("query"), g.t("vmask"))
    torch.testing.assert_close(ev["start"], g.t("eval_start"), atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(ev["end"], g.t("eval_end"), atol=5e-6, rtol=1e-5)


def test_losses(golden):
    g = golden("losses")
    s, e = g.t("start"), g.t("end")
    torch.testing.assert_close(O.span_ground_loss(s, e, g.a["fs"]), g.t("span_ground"), **TOL)
    torch.testing.assert_close(O.bce_loss(g.t("logits"), g.t("labels"), g.t("mask")), g.t("bce"), **TOL)
    kl = O.matching_kl_divergence(O.masked_softmax(g.t("logits"), g.t("labels")),
                                  O.masked_softmax(g.t("logits2"), g.t("labels2")),
                                  g.a["fs"].tolist(), g.a["fs2"].tolist())
    torch.testing.assert_close(kl, g.t("kl"), **TOL)
    torch.testing.assert_close(O.temporal_order_discrimination_loss(g.t("od"), g.t("pd")), g.t("tod"), **TOL)
    pred, score = O.span_pred(s, e)
    assert torch.equal(pred, g.t("pred"))
    torch.testing.assert_close(score, g.t("score"), **TOL)
    torch.testing.assert_close(O.compute_mean_iou(pred.float(), g.t("seg2")), g.t("miou"), **TOL)


def test_aug_and_masks(golden):
    g = golden("aug")
    for i in range(len(g.a["fs"])):
        nf, n, nv = O.gt_moment_translate(g.a["fs"][i].tolist(), int(g.a["nfeats"][i]), g.a["video"][i], int(g.a["pos"][i]))
        assert list(nf) == g.a["new_fs"][i].tolist(), i
        np.testing.assert_array_equal(nv, g.a["new_video"][i])
    for b, m in zip(g.a["seqmask_b"], g.a["seqmask"]):
        np.testing.assert_array_equal(O.sequence_mask(10, b.tolist()), m)
    # the demo the reference's __main__ block prints (data_augment.py:202-225)
    v = np.zeros((1, 40, 1)); v[0, :, 0] = np.arange(40)
    nf, _, nv = O.gt_moment_translate([3, 6], 12, v, 5)
    assert nf == [5, 8] and nv[0, :13, 0].tolist() == [0, 1, 2, 7, 8, 3, 4, 5, 6, 9, 10, 11, 0]


@pytest.mark.parametrize("name", ["charades_cd", "anet_cd"])
def test_iou_scorer_known_answers(golden, name):
    """The committed prediction files re-scored: must equal the reference scorer's output and the
    numbers in the authors' logs (grounding/ckp/*/test.log:84,87)."""
    g = golden("iou")
    miou, recall = O.retrieval_eval(g.a[name + "_pred"], g.a[name + "_gt"])
    got = np.array([miou] + recall)
    np.testing.assert_allclose(got, g.a[name + "_expected"], atol=1e-9)
    np.testing.assert_allclose(got, g.a[name + "_logged"], atol=1e-9)


def test_state_dict_contract_file():
    with open(os.path.join(os.path.dirname(__file__), "golden", "gmd_state_dict_contract.json")) as f:
        c = json.load(f)
    assert len(c) == 80 and sum(int(np.prod(s)) for s in c.values()) == 13847233
