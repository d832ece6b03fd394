"""CPU oracle for the cross-modal matching hot path of haojc/ShufflingVideosForTSG.

TEST INFRASTRUCTURE ONLY.  Nothing under ``shufflingvideosfortsg_amd/`` imports this file.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the reported CPU baseline -- never as the product path.

It is a *functional* restatement (explicit weights in, tensors out; vectorised torch on CPU,
autograd-differentiable) of the reference's ``nn.Module`` graph.  Every function cites the
reference ``file:line`` it follows (paths relative to the reference root).  Weights are passed
as a ``state_dict``-style mapping that uses the reference's parameter names,
so ``oracle.baseline_forward(product_model.state_dict(), ...)`` checks a product model directly.

PARITY PINNING: ``oracle/make_golden.py`` imports the real reference from ``/root/reference``
(in the build container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (fp32, <=1e-6 abs) and the IoU scorer against
the numbers the reference logged for its committed prediction files.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Mapping[str, Tensor]


def _sub(sd: Weights, prefix: str) -> Dict[str, Tensor]:
    """View of ``sd`` restricted to keys under ``prefix`` (prefix stripped)."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------------------
# networks/attention.py
# --------------------------------------------------------------------------------------

def mask_logits(x: Tensor, mask: Tensor, mask_value: float = -1e30) -> Tensor:
    """grounding/model/networks/attention.py:129-133 -- ``x*m + v*(1-m)``."""
    m = mask.to(x.dtype)
    if m.dim() == x.dim() - 1:
        m = m.unsqueeze(-1).expand(-1, -1, x.size(-1))
    return x * m + mask_value * (1.0 - m)


def masked_softmax(vec: Tensor, mask: Tensor, dim: int = 1, epsilon: float = 1e-4) -> Tensor:
    """attention.py:123-127 -- no max-subtraction, ``+epsilon`` in the denominator."""
    e = torch.exp(vec) * mask.float()
    return e / (e.sum(dim, keepdim=True) + epsilon)


def positional_encodings(T: int, D: int, dtype=torch.float32) -> Tensor:
    """attention.py:16-35 -- even channel c: sin(pos/10000^(c/D)); odd: cos(pos/10000^((c-1)/D)).

    The reference fills a [T, D] float32 tensor channel by channel from float32 positions and a
    Python-float divisor; the vectorised form below keeps the same operand types (float32
    position divided by a float64->float32-rounded scalar is what ``tensor / python_float`` does).
    """
    pos = torch.arange(0, T).float()
    enc = torch.zeros(T, D)
    for c in range(D):
        if c % 2 == 0:
            enc[:, c] = torch.sin(pos / 10000 ** (c / D))
        else:
            enc[:, c] = torch.cos(pos / 10000 ** ((c - 1) / D))
    return enc.to(dtype)


def scdm_attention(video: Tensor, sent: Tensor, Ws: Tensor, Wa: Tensor, ba: Tensor, w: Tensor,
                   return_p: bool = False):
    """attention.py:109-121 (``SCDM_Attention.forward``).

    P[b,t,n] = w . tanh(W_s sent[b,n] + W_a video[b,t] + b_a); softmax over the N words
    (no word mask); C = P @ sent.  ``w`` is the [1,H] weight of the bias-free ``self.w``.
    """
    s = F.linear(sent, Ws)                      # [B,N,H]   :112
    a = F.linear(video, Wa, ba)                 # [B,T,H]   :113
    h = torch.tanh(a.unsqueeze(2) + s.unsqueeze(1))   # [B,T,N,H]  :116 (loop over n vectorised)
    e = torch.matmul(h, w.reshape(-1))          # [B,T,N]
    P = torch.softmax(e, dim=-1)                # :118
    C = torch.bmm(P, sent)                      # :119
    return (C, P) if return_p else C


def scdm_core(a: Tensor, s: Tensor, w: Tensor, sent: Tensor):
    """The fused-kernel part of ``scdm_attention`` on *projected* inputs:
    a=[B,T,H] (= W_a video + b), s=[B,N,H] (= W_s sent), w=[H], sent=[B,N,Ds] -> (C, P)."""
    e = torch.matmul(torch.tanh(a.unsqueeze(2) + s.unsqueeze(1)), w.reshape(-1))
    P = torch.softmax(e, dim=-1)
    return torch.bmm(P, sent), P


def attention(q: Tensor, k: Tensor, v: Tensor, d_key: int, causal: bool = False):
    """attention.py:45-55 (``Attention.forward``; dropout p=0 / eval).

    scale = sqrt(d_key) where d_key is whatever the constructor was given (attention.py:41).
    Causal: subtract 1e10*triu(ones(Tk,Tk),1) from the raw dot products *before* the division
    (attention.py:47-52; done on ``.data`` in the reference, i.e. not seen by autograd -- a
    constant shift, so gradients are identical).  Returns (out, A, A_softmax).
    """
    dots = torch.matmul(q, k.transpose(1, 2))
    if q.dim() == 3 and causal:
        tri = torch.ones(k.size(1), k.size(1)).triu(1) * 1e10
        dots = dots - tri.unsqueeze(0).to(dots.dtype)
    A = dots / math.sqrt(d_key)
    S = F.softmax(A, dim=-1)
    return torch.matmul(S, v), A, S


def multihead(q: Tensor, k: Tensor, v: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor,
              n_heads: int, causal: bool = False, return_maps: bool = False):
    """attention.py:71-97 (``MultiHead.forward`` / ``A_forward``).

    Bias-free projections, ``chunk(n_heads,-1)``, per-head ``attention`` with scale
    sqrt(d_key) of the FULL width (attention.py:61 builds ``Attention(d_key, ...)``),
    concat, ``wo``.  With ``return_maps``: also sum_h A_h and sum_h softmax_h (A_forward :95-96).
    """
    d_key = wq.shape[0]
    Q, K, V = F.linear(q, wq), F.linear(k, wk), F.linear(v, wv)
    outs, As, Ss = [], [], []
    for qh, kh, vh in zip(Q.chunk(n_heads, -1), K.chunk(n_heads, -1), V.chunk(n_heads, -1)):
        o, A, S = attention(qh, kh, vh, d_key, causal)
        outs.append(o); As.append(A); Ss.append(S)
    out = F.linear(torch.cat(outs, -1), wo)
    if return_maps:
        return out, torch.stack(As).sum(0), torch.stack(Ss).sum(0)
    return out


def mha_core(Q: Tensor, K: Tensor, V: Tensor, n_heads: int, scale_dim: int, causal: bool = False):
    """Fused-kernel part of ``multihead`` on projected inputs (App. B K2):
    returns (O [B,Tq,dv], A_sum [B,Tq,Tk], S_sum [B,Tq,Tk])."""
    outs, As, Ss = [], [], []
    for qh, kh, vh in zip(Q.chunk(n_heads, -1), K.chunk(n_heads, -1), V.chunk(n_heads, -1)):
        o, A, S = attention(qh, kh, vh, scale_dim, causal)
        outs.append(o); As.append(A); Ss.append(S)
    return torch.cat(outs, -1), torch.stack(As).sum(0), torch.stack(Ss).sum(0)


# --------------------------------------------------------------------------------------
# networks/RNN.py
# --------------------------------------------------------------------------------------

def bilstm(x: Tensor, p: Weights, num_layers: int = 2):
    """grounding/model/networks/RNN.py:34-48 (``BiLSTM.forward``), eval mode (no inter-layer dropout).

    ``nn.LSTM(batch_first=True, bidirectional=True)`` from zero (h0, c0), written out as the cell
    recurrence (gate order i,f,g,o; ``weight_ih_l{k}[_reverse]`` naming) so that it is an
    independent restatement rather than a call into the same ATen kernel.
    Returns (out [B,L,2h], hn [2*layers,B,h], cn).
    """
    B, L, _ = x.shape
    inp = x
    hn: List[Tensor] = []
    cn: List[Tensor] = []
    for layer in range(num_layers):
        dir_out = []
        for suffix, reverse in (("", False), ("_reverse", True)):
            w_ih = p[f"lstm.weight_ih_l{layer}{suffix}"]
            w_hh = p[f"lstm.weight_hh_l{layer}{suffix}"]
            bias = p[f"lstm.bias_ih_l{layer}{suffix}"] + p[f"lstm.bias_hh_l{layer}{suffix}"]
            hsz = w_hh.shape[1]
            gx = F.linear(inp, w_ih, bias)            # [B,L,4h] input GEMM hoisted out of the loop
            h = x.new_zeros(B, hsz)
            c = x.new_zeros(B, hsz)
            steps = range(L - 1, -1, -1) if reverse else range(L)
            outs = [None] * L
            for t in steps:
                g = gx[:, t] + F.linear(h, w_hh)
                i, f, gg, o = g.chunk(4, -1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                outs[t] = h
            dir_out.append(torch.stack(outs, 1))
            hn.append(h); cn.append(c)
        inp = torch.cat(dir_out, -1)
    return inp, torch.stack(hn, 0), torch.stack(cn, 0)


# --------------------------------------------------------------------------------------
# components
# --------------------------------------------------------------------------------------

def sentence_encoder(query: Tensor, p: Weights, num_layers: int = 2):
    """components/SentenceEncoder.py:28-32: Linear(300,300) -> BiLSTM -> (word_feat, cat(hn[-2],hn[-1]))."""
    emb = F.linear(query, p["word_embed.weight"], p["word_embed.bias"])
    out, hn, _ = bilstm(emb, _sub(p, "rnn_cell."), num_layers)
    return out, torch.cat((hn[-2], hn[-1]), -1)


def recalibration_layer(video: Tensor, word: Tensor, p: Weights, num_layers: int = 2):
    """components/VideoEncoder.py:61-74 (``rnn_recalibration_layer.forward``, ca_activ='sigmoid' :84)."""
    r, _, _ = bilstm(video, _sub(p, "rnn_cell."), num_layers)
    C = scdm_attention(r, word, p["attention.W_s.weight"], p["attention.W_a.weight"],
                       p["attention.W_a.bias"], p["attention.w.weight"])
    gate = torch.sigmoid(F.linear(C, p["sent_linear.weight"], p["sent_linear.bias"]))
    return r * gate


def query_aware_encoder(video: Tensor, word: Tensor, p: Weights, nblocks: int = 2, num_layers: int = 2):
    """components/VideoEncoder.py:98-114: nblocks chained recalibration layers sharing ``word``; LayerNorm(eps 1e-5)."""
    x = video
    for i in range(nblocks):
        x = recalibration_layer(x, word, _sub(p, f"blocks.{i}."), num_layers)
    return F.layer_norm(x, (x.size(-1),), p["norm.weight"], p["norm.bias"], 1e-5)


def video_sentence_concat(video: Tensor, sent: Tensor) -> Tensor:
    """components/CrossModalInteraction.py:44-47."""
    return torch.cat([video, sent.unsqueeze(1).expand(-1, video.size(1), -1)], dim=-1)


def mlp_predictor(x: Tensor, p: Weights, v_mask: Optional[Tensor] = None):
    """components/SpanPredictor.py:71-85 (``MLP_predictor.forward``)."""
    def branch(n):
        z = torch.tanh(F.linear(x, p[f"{n}_mlp_1.weight"], p[f"{n}_mlp_1.bias"]))
        l = F.linear(z, p[f"{n}_mlp_2.weight"], p[f"{n}_mlp_2.bias"]).squeeze(2)
        if v_mask is not None:
            l = mask_logits(l, v_mask)
        return torch.softmax(l, dim=1)
    return branch("start"), branch("end")


def self_attention_predictor(x: Tensor, p: Weights, n_heads: int, position_encoding: bool):
    """components/SpanPredictor.py:256-266 (``Self_Attention_predictor.forward``; eval, dropout off)."""
    if position_encoding:
        x = x + positional_encodings(x.size(1), x.size(2)).to(x.dtype)
    def branch(n):
        f = multihead(x, x, x, p[f"{n}_selfattn.wq.weight"], p[f"{n}_selfattn.wk.weight"],
                      p[f"{n}_selfattn.wv.weight"], p[f"{n}_selfattn.wo.weight"], n_heads)
        return torch.softmax(F.linear(f, p[f"{n}_fc.weight"], p[f"{n}_fc.bias"]).squeeze(2), dim=1)
    return branch("start"), branch("end")


def csmm(video: Tensor, sent: Tensor, p: Weights) -> Tensor:
    """components/DistributionAlign.py:112-118 with the 'concat' / NoTemporal / TwoLayerdMLP(relu)
    selections the selectors always return (:17-40): raw per-clip logits [B,T]."""
    x = video_sentence_concat(video, sent)
    hdn = torch.relu(F.linear(x, p["predict.predict.0.weight"], p["predict.predict.0.bias"]))
    return F.linear(hdn, p["predict.predict.2.weight"], p["predict.predict.2.bias"]).squeeze(2)


def moment_pooling(feat: Tensor, target: Tensor, fore: Tensor, back: Tensor, p: Weights) -> Tensor:
    """components/TemporalOrderDiscriminator.py:29-46 (eval: dropout off)."""
    def avg(m):
        return mask_logits(feat, m, 0.0).sum(1) / (m.sum(1, keepdim=True) + 1e-6)
    tgt, fo, ba = avg(target), avg(fore), avg(back)
    ctx = lambda z: torch.relu(F.linear(z, p["foreback_context.0.weight"], p["foreback_context.0.bias"]))
    ff = ctx(torch.cat((fo, tgt), -1))
    bf = ctx(torch.cat((tgt, ba), -1))
    return F.linear(torch.cat((tgt, ff, bf), -1),
                    p["fc_classifier_domain_video.0.weight"], p["fc_classifier_domain_video.0.bias"])


# --------------------------------------------------------------------------------------
# model assembly
# --------------------------------------------------------------------------------------

def baseline_forward(sd: Weights, video: Tensor, query: Tensor, video_mask: Optional[Tensor] = None,
                     use_mask: bool = False, num_layers: int = 2):
    """grounding/model/Baseline.py:63-95 (forward == eval_forward :97-127). query_mask is ignored there."""
    word, sent = sentence_encoder(query, _sub(sd, "sentence_encoder."), num_layers)
    frame = query_aware_encoder(video, word, _sub(sd, "video_encoder."), 2, num_layers)
    cross = video_sentence_concat(frame, sent)
    s, e = mlp_predictor(cross, _sub(sd, "span_predictor.predictor."), video_mask if use_mask else None)
    return {"start": s, "end": e}


def gmd_forward(sd: Weights, query, ori_video, ori_mask, pseudo_video, pseudo_mask,
                ori_t, ori_f, ori_b, ps_t, ps_f, ps_b, use_mask: bool = False, num_layers: int = 2):
    """grounding/model/SpanGroundMatchDisc.py:60-100 (``GMD.forward``)."""
    word, sent = sentence_encoder(query, _sub(sd, "sentence_encoder."), num_layers)
    ve = _sub(sd, "video_encoder.")
    of = query_aware_encoder(ori_video, word, ve, 2, num_layers)
    pf = query_aware_encoder(pseudo_video, word, ve, 2, num_layers)
    cross = video_sentence_concat(of, sent)
    cs = _sub(sd, "csmm.")
    om, pm = csmm(of, sent, cs), csmm(pf, sent, cs)
    gated = om.unsqueeze(2) * cross                               # :86 raw logits gate
    s, e = mlp_predictor(gated, _sub(sd, "span_predictor.predictor."), ori_mask if use_mask else None)
    td = _sub(sd, "tod.")
    return ({"start": s, "end": e}, om, pm,
            moment_pooling(of, ori_t, ori_f, ori_b, td), moment_pooling(pf, ps_t, ps_f, ps_b, td))


def gmd_eval_forward(sd: Weights, video, query, video_mask=None, use_mask: bool = False, num_layers: int = 2):
    """SpanGroundMatchDisc.py:102-129 (``GMD.eval_forward``)."""
    word, sent = sentence_encoder(query, _sub(sd, "sentence_encoder."), num_layers)
    frame = query_aware_encoder(video, word, _sub(sd, "video_encoder."), 2, num_layers)
    cross = video_sentence_concat(frame, sent)
    gated = csmm(frame, sent, _sub(sd, "csmm.")).unsqueeze(2) * cross
    s, e = mlp_predictor(gated, _sub(sd, "span_predictor.predictor."), video_mask if use_mask else None)
    return {"start": s, "end": e}


# --------------------------------------------------------------------------------------
# loss.py
# --------------------------------------------------------------------------------------

def span_ground_loss(start_prob: Tensor, end_prob: Tensor, framestamps) -> Tensor:
    """grounding/loss.py:22-28: mean_b(-log p_s[gt_s] - log p_e[gt_e]) (no epsilon)."""
    fs = torch.as_tensor(np.asarray(framestamps), dtype=torch.long)
    idx = torch.arange(start_prob.size(0))
    return (-torch.log(start_prob[idx, fs[:, 0]]) - torch.log(end_prob[idx, fs[:, 1]])).sum() / len(fs)


def bce_loss(logits: Tensor, labels: Tensor, mask: Tensor) -> Tensor:
    """loss.py:30-36."""
    per = F.binary_cross_entropy_with_logits(logits, labels.type_as(logits), reduction="none")
    m = mask.type_as(logits)
    return (per * m).sum() / (m.sum() + 1e-4)


def kl_divergence(p1: Tensor, p2: Tensor, epsilon: float = 1e-4) -> Tensor:
    """loss.py:38-40."""
    return torch.sum(p1 * torch.log((p1 + epsilon) / (p2 + epsilon)), dim=-1)


def matching_kl_divergence(prob1: Tensor, prob2: Tensor, fs1, fs2) -> Tensor:
    """loss.py:42-51: KL between the GT-moment slice of the original and of the translated video."""
    assert len(fs1) == len(fs2)
    loss = 0
    for i in range(len(fs1)):
        s1, e1 = fs1[i]; s2, e2 = fs2[i]
        loss = loss + kl_divergence(prob1[i][s1:e1 + 1], prob2[i][s2:e2 + 1])
    return loss / len(fs1)


def temporal_order_discrimination_loss(ori_logit: Tensor, pseudo_logit: Tensor) -> Tensor:
    """loss.py:6-20 with ``criterion_domain = nn.CrossEntropyLoss()`` (train.py:387): label 0 = original."""
    o = ori_logit.view(-1, ori_logit.size(-1)); p = pseudo_logit.view(-1, pseudo_logit.size(-1))
    lab = torch.cat((torch.zeros(o.size(0)), torch.ones(p.size(0)))).long()
    return F.cross_entropy(torch.cat((o, p), 0), lab)


def span_pred(start_prob: Tensor, end_prob: Tensor):
    """loss.py:53-70: argmax_{i<=j} (start_i + end_j), zeros below the diagonal take part in the max
    (``triu(0)`` zero-fills), first maximum wins.  Returns (int64 [B,2], score [B])."""
    B, T = start_prob.shape
    m = (start_prob.unsqueeze(2) + end_prob.unsqueeze(1)).triu(0)
    row_max, row_idx = m.max(dim=2)
    best, col = row_max.max(dim=1)
    end = row_idx[torch.arange(B), col]
    return torch.stack((col, end), -1), best


def compute_mean_iou(seg1: Tensor, seg2: Tensor) -> Tensor:
    """loss.py:72-91: union = max_end - min_beg, +1e-4."""
    s1, e1 = seg1[:, 0], seg1[:, 1]; s2, e2 = seg2[:, 0], seg2[:, 1]
    inter = (torch.minimum(e1, e2) - torch.maximum(s1, s2)).clamp(min=0)
    union = torch.maximum(e1, e2) - torch.minimum(s1, s2)
    return (inter / (union + 1e-4)).mean()


def gmd_losses(out, ori_mask, pseudo_mask, ori_gt, pseudo_gt, lam=(1.0, 1.0, 1.0)):
    """train.py:142-165: total loss and its four parts for one GMD step."""
    span, om, pm, od, pd = out
    lg = span_ground_loss(span["start"], span["end"], ori_gt["framestps"])
    l1 = lam[0] * (bce_loss(om, ori_gt["temporal_labels"], ori_mask)
                   + bce_loss(pm, pseudo_gt["temporal_labels"], pseudo_mask))
    l2 = lam[1] * matching_kl_divergence(masked_softmax(om, ori_gt["temporal_labels"]),
                                         masked_softmax(pm, pseudo_gt["temporal_labels"]),
                                         ori_gt["framestps"], pseudo_gt["framestps"])
    ld = temporal_order_discrimination_loss(od, pd)
    return lg + l1 + l2 + lam[2] * ld, (lg, l1, l2, ld)


# --------------------------------------------------------------------------------------
# dataset helpers + IoU scorer
# --------------------------------------------------------------------------------------

def sequence_mask(max_len: int, boundary: Sequence[int]) -> np.ndarray:
    """grounding/dataset/charades.py:12-18 -- inclusive [st, et], clipped."""
    st, et = boundary
    m = np.zeros([max_len], dtype=np.int32)
    m[max(0, st):min(et, max_len - 1) + 1] = 1
    return m


def gt_moment_translate(framestps, nfeats: int, video_feat: np.ndarray, cropin_start: int):
    """grounding/dataset/data_augment.py:135-156 with the random insert position made an argument.

    ``video_feat`` is [1,T,D].  Cut frames [s,e] out, close the gap, re-insert the moment in front
    of position ``cropin_start`` (0..nfeats-len) of the gap-closed sequence; nfeats unchanged.
    """
    s, e = framestps
    n = e - s + 1
    if n <= 1 or n >= nfeats:
        return list(framestps), nfeats, video_feat
    T = video_feat.shape[1]
    rest = np.concatenate([video_feat[0, :s], video_feat[0, e + 1:nfeats]], 0)      # nfeats-n rows
    seq = np.concatenate([rest[:cropin_start], video_feat[0, s:e + 1], rest[cropin_start:]], 0)
    out = np.zeros(video_feat.shape) + 0.0
    out[0, :min(T, seq.shape[0])] = seq[:T]
    return [cropin_start, cropin_start + n - 1], nfeats, out


def segment_iou(target: np.ndarray, cand: np.ndarray) -> np.ndarray:
    """grounding/IoU_eval.py:8-34: union = len1 + len2 - inter, +1e-4."""
    inter = (np.minimum(target[1], cand[:, 1]) - np.maximum(target[0], cand[:, 0])).clip(0)
    union = (cand[:, 1] - cand[:, 0]) + (target[1] - target[0]) - inter
    return inter.astype(float) / (union + 1e-4)


def retrieval_eval(pred: np.ndarray, gt: np.ndarray, thresholds=(0.1, 0.3, 0.5, 0.7, 0.9)):
    """IoU_eval.py:94-153 vectorised: every (video, sentence-index) is its own group holding one
    proposal (:71-79), so R@1 = mean(iou > thr) (strict, :136) and mIoU = round(mean*100, 2) (:145).
    The accumulator is zero-initialised (the reference's ``np.empty`` :131 is a bug that can leak
    garbage into the sums).  Returns (mIoU, [R@1 per threshold in percent, rounded to 2 dp])."""
    pred = np.asarray(pred, dtype=np.float64); gt = np.asarray(gt, dtype=np.float64)
    inter = (np.minimum(pred[:, 1], gt[:, 1]) - np.maximum(pred[:, 0], gt[:, 0])).clip(0)
    union = (gt[:, 1] - gt[:, 0]) + (pred[:, 1] - pred[:, 0]) - inter
    iou = inter / (union + 1e-4)
    recall = [round(float((iou > t).sum()) / len(iou) * 100, 2) for t in thresholds]
    return round(float(iou.mean()) * 100, 2), recall


def retrieval_eval_json(data: dict, thresholds=(0.1, 0.3, 0.5, 0.7, 0.9)):
    """Same scorer fed from the submits-JSON schema (IoU_eval.py:60-92)."""
    pred, gt = [], []
    for _, rows in data["results"].items():
        for r in rows:
            pred.append(r["timestamp"]); gt.append(r["gt_timestamp"])
    return retrieval_eval(np.array(pred), np.array(gt), thresholds)
