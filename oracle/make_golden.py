#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference is absent on the GPU box).  The reference is
imported, never copied: this script only draws seeded inputs, calls the reference's classes and
stores inputs / weights / outputs / gradients as data.  Shims applied in THIS process only
(reference files untouched):
  1. ``Tensor.cuda`` / ``Module.cuda`` -> identity (RNN.py:37-38, loss.py:15 hard-code .cuda()).
  2. ``h5py`` stub module (dataset/charades.py:3 imports it; only c3d features use it).
  3. ``Attention`` / ``MultiHead`` are built with ``nn.Module.__init__`` run first, because their
     constructors call the one-argument ``super(Cls).__init__()`` (attention.py:40,60).
  4. ``loss.span_pred`` indexes with a (2,B) numpy array (loss.py:63-66), which torch>=2 rejects;
     the golden for it is produced with the tuple-index form and cross-checked by brute force.

Usage:  python oracle/make_golden.py   (writes tests/golden/, prints a summary)
"""
import contextlib
import io
import json
import logging
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/grounding"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, REF)

from model.networks import attention as ref_att            # noqa: E402
from model.networks.RNN import BiLSTM                       # noqa: E402
from model.components import VideoEncoder, SpanPredictor, CrossModalInteraction  # noqa: E402
from model.components import SentenceEncoder, DistributionAlign, TemporalOrderDiscriminator  # noqa: E402
from model.Baseline import Baseline                         # noqa: E402
from model.SpanGroundMatchDisc import GMD                   # noqa: E402
import loss as ref_loss                                     # noqa: E402
import IoU_eval as ref_iou                                  # noqa: E402
from dataset.charades import Sequence_mask                  # noqa: E402
from dataset import data_augment as ref_aug                 # noqa: E402

torch.set_num_threads(4)
LOG = logging.getLogger("golden")


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} kB  ({len(arrs)} arrays)")


def weights(mod, prefix="w."):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def grads(mod, prefix="gw."):
    return {prefix + k: p.grad for k, p in mod.named_parameters()}


class _FixedAttention(ref_att.Attention):
    """Reference ``Attention`` with ``nn.Module.__init__`` run before the reference constructor body
    (whose one-argument ``super(Attention).__init__()`` is a no-op on an unbound super object)."""

    def __init__(self, d_key, drop_ratio, causal):
        nn.Module.__init__(self)
        ref_att.Attention.__mro__[1].__init__(self, d_key, drop_ratio, causal)


class _FixedMultiHead(ref_att.MultiHead):
    def __init__(self, d_key, d_value, n_heads, drop_ratio, causal=False):
        nn.Module.__init__(self)
        ref_att.MultiHead.__mro__[1].__init__(self, d_key, d_value, n_heads, drop_ratio, causal)


# the reference bodies look their own class names up in the module globals (``Attention(...)`` at
# attention.py:61, ``super(Attention)`` at :40): point those names at the constructible subclasses.
ref_att.Attention = _FixedAttention
ref_att.MultiHead = _FixedMultiHead
make_attention = _FixedAttention
make_multihead = _FixedMultiHead


# ---------------------------------------------------------------------------------------
def gen_scdm():
    for tag, (B, T, N, Dv, Ds, H) in {"a": (2, 9, 5, 24, 24, None), "b": (3, 17, 20, 40, 24, 32),
                                      "c": (1, 1, 1, 8, 8, None)}.items():
        torch.manual_seed(10)
        m = ref_att.SCDM_Attention(Dv, Ds, H)
        v = torch.randn(B, T, Dv, requires_grad=True)
        s = torch.randn(B, N, Ds, requires_grad=True)
        C = m(v, s)
        gC = torch.randn_like(C)
        C.backward(gC)
        save(f"scdm_{tag}", video=v, sent=s, C=C, gC=gC, gvideo=v.grad, gsent=s.grad,
             **weights(m), **grads(m))


def gen_mha():
    cases = {"cross": (2, 9, 5, 32, 4, False), "self": (2, 9, 9, 32, 8, False),
             "causal": (2, 7, 7, 16, 2, True), "onehead": (1, 4, 6, 8, 1, False)}
    for tag, (B, Tq, Tk, d, h, causal) in cases.items():
        torch.manual_seed(20)
        m = make_multihead(d, d, h, 0.0, causal)
        m.eval()
        q = torch.randn(B, Tq, d, requires_grad=True)
        if tag in ("self", "causal"):
            k = v = q
        else:
            k = torch.randn(B, Tk, d, requires_grad=True)
            v = torch.randn(B, Tk, d, requires_grad=True)
        out = m(q, k, v)
        g = torch.randn_like(out)
        out.backward(g)
        extra = {}
        if tag not in ("self", "causal"):
            extra = dict(k=k, v=v, gk=k.grad, gv=v.grad)
        with torch.no_grad():
            out2 = m.A_forward(q, k, v)
        assert torch.equal(out2, out)
        save(f"mha_{tag}", q=q, out=out, g=g, gq=q.grad, A=m.A, A_softmax=m.A_softmax,
             n_heads=np.int64(h), causal=np.int64(causal), **extra, **weights(m), **grads(m))
    # single Attention module, incl. 3-D causal
    torch.manual_seed(21)
    a = make_attention(16, 0.0, True)
    a.eval()
    q, k, v = torch.randn(2, 6, 16), torch.randn(2, 6, 16), torch.randn(2, 6, 16)
    o, A, S = a(q, k, v)
    save("attention_causal", q=q, k=k, v=v, out=o, A=A, S=S)


def gen_posenc():
    x = torch.zeros(1, 13, 10)
    save("posenc", enc=ref_att.positional_encodings_like(x), T=np.int64(13), D=np.int64(10))
    v = torch.randn(3, 7)
    m = (torch.rand(3, 7) > 0.4).int()
    save("mask_helpers", vec=v, mask=m, masked_softmax=ref_att.masked_softmax(v, m),
         mask_logits=ref_att.mask_logits(v, m), mask_logits0=ref_att.mask_logits(v, m, 0.0),
         feat=(f := torch.randn(3, 7, 4)), mask_logits3=ref_att.mask_logits(f, m, 0.0))


def gen_mlp():
    for tag, use_mask in (("nomask", False), ("mask", True)):
        torch.manual_seed(30)
        B, T, D, H = 3, 11, 40, 16
        m = SpanPredictor.MLP_predictor(D, H)
        x = torch.randn(B, T, D, requires_grad=True)
        mask = torch.from_numpy(np.stack([Sequence_mask(T, [0, n]) for n in (10, 5, 7)]))
        s, e = m(x, mask if use_mask else None)
        gs, ge = torch.randn_like(s), torch.randn_like(e)
        (s * gs + e * ge).sum().backward()
        save(f"mlp_{tag}", x=x, mask=mask, start=s, end=e, gs=gs, ge=ge, gx=x.grad,
             **weights(m), **grads(m))
    # VideoSentenceConcat
    cmi = CrossModalInteraction.VideoSentenceConcat(6, 5)
    v, sent = torch.randn(2, 4, 6), torch.randn(2, 5)
    save("concat", video=v, sent=sent, cross=cmi(v, None, sent), cross_dim=np.int64(cmi.cross_dim()))


def gen_selfattn_predictor():
    # Self_Attention_predictor (SpanPredictor.py:244-266) needs the constructor shim for MultiHead
    for tag, pe in (("pe", True), ("nope", False)):
        torch.manual_seed(40)
        D, h = 16, 4
        real = SpanPredictor.MultiHead
        SpanPredictor.MultiHead = make_multihead
        try:
            m = SpanPredictor.Self_Attention_predictor(D, h, pe, 0.0)
        finally:
            SpanPredictor.MultiHead = real
        m.eval()
        x = torch.randn(2, 9, D, requires_grad=True)
        s, e = m(x)
        gs, ge = torch.randn_like(s), torch.randn_like(e)
        (s * gs + e * ge).sum().backward()
        save(f"selfattn_pred_{tag}", x=x, start=s, end=e, gs=gs, ge=ge, gx=x.grad,
             n_heads=np.int64(h), **weights(m), **grads(m))


def gen_bilstm():
    torch.manual_seed(50)
    m = BiLSTM(12, 8, 2, 0.5)
    m.eval()
    x = torch.randn(3, 7, 12, requires_grad=True)
    out, hn, cn = m(x)
    g = torch.randn_like(out)
    gh = torch.randn_like(hn)
    ((out * g).sum() + (hn * gh).sum()).backward()
    save("bilstm", x=x, out=out, hn=hn, cn=cn, g=g, gh=gh, gx=x.grad, **weights(m), **grads(m))


def _sets(in_dim, h, mlp_h, match_h, use_mask=False, drop=0.0):
    video = dict(name="query_aware_encoder", input_dim=in_dim, rnn_hidden_dim=h, rnn_layers=2,
                 rnn_cell="lstm", mask=use_mask, drop_out=drop, T=16, nblocks=2)
    sent = dict(name="rnn", input_dim=300, rnn_hidden_dim=h, rnn_layers=2, rnn_cell="lstm", drop_out=drop)
    ground = dict(cross_name="vs", name="mlp", lstm_hidden_dim=16, mlp_hidden_dim=mlp_h)
    match = dict(cross=dict(name="concat"),
                 temporal=dict(name="none", hidden_dim=256, layers=2, dropout=drop),
                 predict=dict(name="mlp", activation="relu", hidden_dim=match_h))
    return video, sent, ground, match


def _batch(B, T, N, in_dim, seed):
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    video = torch.randn(B, T, in_dim, generator=g)
    query = torch.randn(B, N, 300, generator=g) * 0.4
    nfeats = rs.randint(T // 2, T + 1, size=B)
    vmask = torch.from_numpy(np.stack([Sequence_mask(T, [0, int(n)]) for n in nfeats]))
    qmask = torch.ones(B, N, dtype=torch.int32)
    fs = []
    for n in nfeats:
        s = rs.randint(0, n - 1); e = rs.randint(s + 1, n)
        fs.append([int(s), int(e)])
    return video, query, vmask, qmask, nfeats, fs


def gen_qave_and_models():
    # QueryAwareEncoder alone
    torch.manual_seed(60)
    vs, ss, gs_, ms = _sets(20, 8, 12, 16)
    vs["query_dim"] = 16
    enc = VideoEncoder.QueryAwareEncoder(vs, LOG)
    enc.eval()
    v = torch.randn(2, 9, 20, requires_grad=True)
    w = torch.randn(2, 5, 16, requires_grad=True)
    o = enc(v, w)
    g = torch.randn_like(o)
    o.backward(g)
    save("qave", video=v, word=w, out=o, g=g, gvideo=v.grad, gword=w.grad, **weights(enc), **grads(enc))

    # Baseline, small dims, +/- mask
    for tag, use_mask in (("nomask", False), ("mask", True)):
        torch.manual_seed(61)
        vs, ss, gs_, ms = _sets(24, 8, 12, 16, use_mask)
        model = Baseline(vs, ss, gs_, ms, LOG, 0.0)
        model.eval()
        video, query, vmask, qmask, nfeats, fs = _batch(3, 12, 6, 24, 5)
        out = model(video, query, vmask, qmask)
        loss = ref_loss.span_ground_loss(out["start"], out["end"], fs)
        loss.backward()
        pred, score = _span_pred_fixed(out["start"].detach(), out["end"].detach())
        save(f"baseline_{tag}", video=video, query=query, vmask=vmask, framestps=np.array(fs),
             start=out["start"], end=out["end"], loss=loss, pred=pred, score=score,
             **weights(model), **grads(model))

    # GMD, small dims
    torch.manual_seed(62)
    vs, ss, gs_, ms = _sets(24, 8, 12, 16)
    model = GMD(vs, ss, gs_, ms, LOG, 0.0)
    model.eval()   # MomentPooling.dropout(p=.5) off
    B, T, N = 3, 12, 6
    video, query, vmask, qmask, nfeats, fs = _batch(B, T, N, 24, 6)
    aug = ref_aug.DataAugmentForTSG(0, 1, "gt_translate")
    pv, pfs = [], []
    real_randint = random.randint
    for b in range(B):
        wo = int(nfeats[b]) - (fs[b][1] - fs[b][0] + 1)
        random.randint = lambda lo, hi, _w=wo, _b=b: (_b * 3 + 1) % (_w + 1)
        nf, _, nv = aug.gt_moment_translate(fs[b], int(nfeats[b]), video[b:b + 1].double().numpy())
        pv.append(torch.from_numpy(nv[0]).float()); pfs.append([int(nf[0]), int(nf[1])])
    random.randint = real_randint
    pvideo = torch.stack(pv)

    def labels(fsl):
        t = np.stack([Sequence_mask(T, f) for f in fsl])
        fo = np.stack([Sequence_mask(T, [0, f[0]]) for f in fsl])
        ba = np.stack([Sequence_mask(T, [f[1], int(n)]) for f, n in zip(fsl, nfeats)])
        return torch.from_numpy(t), torch.from_numpy(fo), torch.from_numpy(ba)
    ot, of, ob = labels(fs)
    pt, pf, pb = labels(pfs)
    out = model(query, qmask, video, vmask, pvideo, vmask, ot, of, ob, pt, pf, pb)
    span, om, pm, od, pd = out
    lg = ref_loss.span_ground_loss(span["start"], span["end"], fs)
    l1 = ref_loss.BCE_loss(om, ot, vmask) + ref_loss.BCE_loss(pm, pt, vmask)
    l2 = ref_loss.matching_KL_divergence(ref_att.masked_softmax(om, ot), ref_att.masked_softmax(pm, pt), fs, pfs)
    ld = ref_loss.temporal_order_discrimination_loss(od, pd, nn.CrossEntropyLoss())
    loss = lg + l1 + l2 + ld
    loss.backward()
    with torch.no_grad():
        ev = model.eval_forward(video, query, vmask, qmask)
    save("gmd", video=video, pvideo=pvideo, query=query, vmask=vmask, framestps=np.array(fs),
         pframestps=np.array(pfs), ot=ot, of=of, ob=ob, pt=pt, pf=pf, pb=pb,
         start=span["start"], end=span["end"], om=om, pm=pm, od=od, pd=pd,
         lg=lg, l1=l1, l2=l2, ld=ld, loss=loss, eval_start=ev["start"], eval_end=ev["end"],
         **weights(model), **grads(model))

    # state_dict contract at the default dims (names + shapes only; Appendix A)
    torch.manual_seed(0)
    vs, ss, gs_, ms = _sets(1024, 256, 256, 1024)
    model = GMD(vs, ss, gs_, ms, LOG, 0.5)
    contract = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "gmd_state_dict_contract.json"), "w") as f:
        json.dump(contract, f, indent=0)
    print(f"  gmd_state_dict_contract.json ({len(contract)} tensors, "
          f"{sum(int(np.prod(s)) for s in contract.values())} params)")


def _span_pred_fixed(start, end):
    """loss.span_pred (loss.py:53-70) with tuple indexing; checked against brute force."""
    B, T = start.shape
    sm = start.unsqueeze(-1).expand(B, T, T)
    em = end.unsqueeze(-1).expand(B, T, T).permute(0, 2, 1)
    pm = (sm + em).triu(diagonal=0)
    row_max, row_idx = pm.max(dim=2)
    best, col = row_max.max(dim=1)
    endi = row_idx[torch.arange(B), col]
    for b in range(B):     # brute force
        bv, bi = -1.0, None
        for i in range(T):
            for j in range(i, T):
                val = float(start[b, i] + end[b, j])
                if val > bv:
                    bv, bi = val, (i, j)
        assert bi == (int(col[b]), int(endi[b])), (bi, col[b], endi[b])
    return torch.stack((col, endi), -1), best


def gen_losses():
    torch.manual_seed(70)
    B, T = 4, 10
    s = torch.softmax(torch.randn(B, T), 1)
    e = torch.softmax(torch.randn(B, T), 1)
    fs = [[1, 4], [0, 9], [3, 3], [2, 7]]
    fs2 = [[0, 3], [0, 9], [5, 5], [4, 9]]
    logits, logits2 = torch.randn(B, T), torch.randn(B, T)
    labels = torch.from_numpy(np.stack([Sequence_mask(T, f) for f in fs]))
    labels2 = torch.from_numpy(np.stack([Sequence_mask(T, f) for f in fs2]))
    mask = torch.from_numpy(np.stack([Sequence_mask(T, [0, n]) for n in (9, 10, 6, 8)]))
    pred, score = _span_pred_fixed(s, e)
    seg1 = pred.float()
    seg2 = torch.tensor([[1.5, 4.2], [0., 8.], [6., 7.], [2., 7.5]])
    od, pd = torch.randn(B, 2), torch.randn(B, 2)
    save("losses", start=s, end=e, fs=np.array(fs), fs2=np.array(fs2), logits=logits, logits2=logits2,
         labels=labels, labels2=labels2, mask=mask,
         span_ground=ref_loss.span_ground_loss(s, e, fs),
         bce=ref_loss.BCE_loss(logits, labels, mask),
         kl=ref_loss.matching_KL_divergence(ref_att.masked_softmax(logits, labels),
                                            ref_att.masked_softmax(logits2, labels2), fs, fs2),
         tod=ref_loss.temporal_order_discrimination_loss(od, pd, nn.CrossEntropyLoss()), od=od, pd=pd,
         pred=pred, score=score, seg2=seg2, miou=ref_loss.compute_mean_iou(seg1, seg2))


def gen_aug():
    aug = ref_aug.DataAugmentForTSG(0, 1, "gt_translate")
    real = random.randint
    cases = []
    T, D = 16, 3
    rs = np.random.RandomState(3)
    for i, (fs, nfeats, pos) in enumerate([([3, 6], 12, 5), ([0, 2], 16, 13), ([10, 15], 16, 0),
                                            ([4, 4], 12, 2), ([0, 11], 12, 0), ([5, 9], 10, 5),
                                            ([1, 8], 9, 1)]):
        v = np.zeros((1, T, D)); v[0, :nfeats] = rs.randn(nfeats, D)
        random.randint = lambda lo, hi, _p=pos: _p
        nf, n2, nv = aug.gt_moment_translate(list(fs), nfeats, v.copy())
        cases.append((fs, nfeats, pos, v, nf, nv))
    random.randint = real
    save("aug", fs=np.array([c[0] for c in cases]), nfeats=np.array([c[1] for c in cases]),
         pos=np.array([c[2] for c in cases]), video=np.stack([c[3] for c in cases]),
         new_fs=np.array([c[4] for c in cases]), new_video=np.stack([c[5] for c in cases]),
         seqmask=np.stack([Sequence_mask(10, b) for b in ([0, 4], [3, 12], [-2, 2], [9, 9], [0, 10])]),
         seqmask_b=np.array([[0, 4], [3, 12], [-2, 2], [9, 9], [0, 10]]))


def gen_iou():
    res = {}
    for name, path in {"charades_cd": "ckp/charades_cd/prediction_results_test_ood.json",
                       "anet_cd": "ckp/anet_cd/prediction_results_test_ood.json"}.items():
        full = os.path.join(REF, path)
        data = json.load(open(full))
        pred, gt = [], []
        for _, rows in data["results"].items():
            for r in rows:
                pred.append(r["timestamp"]); gt.append(r["gt_timestamp"])
        # the reference scorer, with its uninitialised accumulator (IoU_eval.py:131) zeroed
        real_empty = np.empty
        ref_iou.np.empty = lambda shape, *a, **k: np.zeros(shape, *a, **k)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_iou.retrieval_eval(full)
        ref_iou.np.empty = real_empty
        line = [l for l in buf.getvalue().splitlines() if l.startswith("1 ")][0].split("\t")
        nums = [float(x) for x in line[1:] if x.strip()]
        print("   ", name, "reference scorer ->", nums)
        res[name + "_pred"] = np.array(pred, dtype=np.float64)
        res[name + "_gt"] = np.array(gt, dtype=np.float64)
        res[name + "_expected"] = np.array(nums)     # [mIoU, R@.1, .3, .5, .7, .9]
    # numbers the authors' logs hold (test.log:84 / :87)
    res["charades_cd_logged"] = np.array([44.28, 75.35, 63.85, 46.84, 27.47, 6.64])
    res["anet_cd_logged"] = np.array([30.21, 66.05, 42.14, 24.58, 13.47, 4.52])
    save("iou", **res)


def gen_config0():
    """BASELINE config 0: QAVE at d=512, B=2, T=32, N=15, real Charades-CD sentences + GloVe rows,
    seeded synthetic i3d features, weights = PyTorch default init under manual_seed(0).  Weights are
    not stored (49 MB); the product model must reproduce them from the same seed (construction-order
    contract), pinned by per-tensor checksums."""
    wordtoix = np.load("/root/reference/data/Charades/words/wordtoix.npy", allow_pickle=True).tolist()
    glove = np.load("/root/reference/data/Charades/words/word_glove_fts_init.npy")
    ann = json.load(open("/root/reference/data/Charades-CD/charades_test_ood.json"))
    items = []
    for vid, a in ann.items():
        for ts, sent in zip(a["timestamps"], a["sentences"]):
            items.append((vid, a["video_duration"], ts, sent))
        if len(items) >= 2:
            break
    items = items[:2]
    N, T = 15, 32
    import string
    q = np.zeros((2, N, 300), dtype=np.float32)
    for b, (_, _, _, sent) in enumerate(items):
        # tokenisation as dataset/charades.py:119-127: punctuation -> ' ', lower, split(' '), keep
        # in-vocabulary words, zero-pad the index list (index 0 is a real embedding row)
        for c in string.punctuation:
            sent = sent.replace(c, " ")
        idx = [wordtoix[w] for w in sent.lower().split(" ") if w in wordtoix][:N]
        idx = idx + [0] * (N - len(idx))
        q[b] = glove[idx].astype(np.float32)
    torch.manual_seed(0)
    vs, ss, gs_, ms = _sets(1024, 256, 256, 1024, False, 0.5)
    model = Baseline(vs, ss, gs_, ms, LOG, 0.5)
    model.eval()
    g = torch.Generator().manual_seed(1234)
    video = torch.randn(2, T, 1024, generator=g)
    vmask = torch.ones(2, T, dtype=torch.int32)
    with torch.no_grad():
        out = model(video, torch.from_numpy(q), vmask, None)
    pred, score = _span_pred_fixed(out["start"], out["end"])
    sd = model.state_dict()
    keys = list(sd.keys())
    save("config0", query=q, video_seed=np.int64(1234), start=out["start"], end=out["end"],
         pred=pred, score=score, keys=np.array(keys),
         wsum=np.array([float(sd[k].double().sum()) for k in keys]),
         wabs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
         sentences=np.array([it[3] for it in items]))


class _TupleIndex:
    """what ``torch.stack((idx, col), 0).numpy()`` was to the torch the reference was written for: an index PAIR"""

    def __init__(self, t):
        self.t = t

    def numpy(self):
        return tuple(self.t[i] for i in range(self.t.shape[0]))


class _TorchProxy:
    """``torch`` as seen by the reference's loss.py, with ``stack`` returning the index-pair wrapper above so that the REAL
    ``loss.span_pred`` (loss.py:53-70) runs unmodified on torch >= 2 (shim 4)."""

    def __getattr__(self, k):
        if k == "stack":
            return lambda ts, dim=0: _TupleIndex(torch.stack(ts, dim=dim))
        return getattr(torch, k)


def _ref_span_pred(start, end):
    real = ref_loss.torch
    ref_loss.torch = _TorchProxy()
    try:
        return ref_loss.span_pred(start, end)
    finally:
        ref_loss.torch = real


def gen_span_pred_cases():
    """The reference's own span_pred on edge cases: random rows, exact ties (first maximum wins), rows whose probabilities are
    exactly zero beyond a mask (the zero-filled lower triangle then takes part in the max), rounding-collapsed sums, T = 1."""
    g = torch.Generator().manual_seed(42)
    cases = {}
    s = torch.softmax(torch.randn(6, 33, generator=g), 1); e = torch.softmax(torch.randn(6, 33, generator=g), 1)
    cases["rand"] = (s, e)
    s = torch.full((3, 8), 0.125); e = torch.full((3, 8), 0.125)                     # all ties
    s[1, 5] = 0.25; e[1, 2] = 0.25; e[1, 6] = 0.25; s[2, 0] = 0.25; s[2, 7] = 0.25
    cases["ties"] = (s, e)
    s = torch.zeros(4, 12); e = torch.zeros(4, 12)                                    # masked tails: exact zeros
    s[0, :5] = torch.softmax(torch.randn(5, generator=g), 0); e[0, :5] = torch.softmax(torch.randn(5, generator=g), 0)
    s[1, 3] = 1.0; e[1, 1] = 1.0                                                      # end mass BEFORE the start
    e[2, 0] = 1.0                                                                     # start all zero
    cases["zeros"] = (s, e)                                                           # row 3: everything zero
    s = torch.tensor([[1.0, 1.0, 0.5], [1.0, 0.25, 0.5]])                             # fl(1 + e_j) collapses: in row 1 all
    e = torch.tensor([[3e-8, 5.9e-8, 6e-8], [6e-8, 7e-8, 8e-8]])                      # three sums are equal, the FIRST wins
    cases["round"] = (s, e)
    cases["t1"] = (torch.tensor([[0.7], [0.0]]), torch.tensor([[0.2], [0.0]]))
    s = torch.softmax(torch.randn(2, 300, generator=g) * 3, 1); e = torch.softmax(torch.randn(2, 300, generator=g) * 3, 1)
    cases["long"] = (s, e)
    out = {}
    for k, (s, e) in cases.items():
        pred, score = _ref_span_pred(s, e)
        p2, s2 = _span_pred_fixed(s, e) if k in ("rand", "long") else (pred, score)   # the tuple-index form agrees
        assert torch.equal(pred, p2) and torch.equal(score, s2)
        out.update({f"{k}.start": s, f"{k}.end": e, f"{k}.pred": pred, f"{k}.score": score})
    save("span_pred_cases", **out)


def gen_pool():
    """Charades pair-mean pooling + zero pad + frame stamps (dataset/charades.py:177-196, generate_video_fts_data), called on
    the real class's method with a stand-in ``self`` holding SAMPLE_LEN; float32 clip features as the i3d .npy files hold."""
    from dataset.charades import CharadesDataSentence as _C          # the class that defines generate_video_fts_data
    rs = np.random.RandomState(5)
    T, D = 12, 8
    self_ = types.SimpleNamespace(SAMPLE_LEN=T)
    raws, outs, nfs, fss, tss = [], [], [], [], []
    for n, ts in ((7, (1.2, 3.9)), (8, (0.0, 2.0)), (1, (0.0, 0.4)), (24, (3.5, 30.0)), (31, (11.9, 12.0)), (23, (0.5, 11.2))):
        raw = rs.randn(n, D).astype(np.float32)
        o, fs, nf = _C.generate_video_fts_data(self_, raw, list(ts), 30.0)
        raws.append(raw); outs.append(o[0]); nfs.append(nf); fss.append(fs); tss.append(ts)
    save("pool", raw=np.concatenate(raws), offsets=np.cumsum([0] + [len(r) for r in raws]).astype(np.int64),
         timestamps=np.array(tss, dtype=np.float64), out=np.stack(outs), nfeats=np.array(nfs, dtype=np.int64),
         framestps=np.array(fss, dtype=np.int64), T=np.int64(T))


if __name__ == "__main__":
    print("writing golden vectors to", os.path.normpath(OUT))
    if len(sys.argv) > 1:                     # only the named generators, e.g. `make_golden.py gen_pool gen_span_pred_cases`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    gen_scdm(); gen_mha(); gen_posenc(); gen_mlp(); gen_selfattn_predictor(); gen_bilstm()
    gen_qave_and_models(); gen_losses(); gen_aug(); gen_iou(); gen_config0(); gen_span_pred_cases(); gen_pool()
