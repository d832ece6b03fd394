"""Losses and span decoding of the grounding path (reference grounding/loss.py), vectorised torch ops
that stay on the tensors' device -- no per-sample Python loops, no ``.cuda()`` / ``.cpu()`` hops.
Same function names and results as the reference (checked against its golden values in tests/)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

DELTA = 1e-4


def _idx(framestamps, device):
    if isinstance(framestamps, torch.Tensor):
        return framestamps.to(device=device, dtype=torch.long)
    return torch.as_tensor(np.asarray(framestamps), dtype=torch.long, device=device)


def span_ground_loss(start_prob, end_prob, framestamps):
    """mean_b( -log p_start[gt_s] - log p_end[gt_e] ), no epsilon (reference loss.py:22-28)."""
    fs = _idx(framestamps, start_prob.device)
    ps = start_prob.gather(1, fs[:, 0:1]).squeeze(1)
    pe = end_prob.gather(1, fs[:, 1:2]).squeeze(1)
    return (-(torch.log(ps) + torch.log(pe))).sum() / fs.size(0)


def BCE_loss(logits, labels, mask):
    """masked mean of BCE-with-logits, denominator mask.sum()+1e-4 (reference loss.py:30-36)."""
    per = F.binary_cross_entropy_with_logits(logits, labels.type_as(logits), reduction='none')
    m = mask.type_as(logits)
    return (per * m).sum() / (m.sum() + DELTA)


def KL_divergence(prob1, prob2, epsilon=1e-4):
    """sum p1 * log((p1+eps)/(p2+eps)) over the last axis (reference loss.py:38-40)."""
    return torch.sum(prob1 * torch.log((prob1 + epsilon) / (prob2 + epsilon)), dim=-1)


def matching_KL_divergence(prob1, prob2, framestps1, framestps2, epsilon=1e-4):
    """mean_b KL(prob1[b, s1:e1+1] || prob2[b, s2:e2+1]); the two slices have equal length (the
    translated moment), reference loss.py:42-51.  Vectorised with a gather over slice offsets."""
    f1, f2 = _idx(framestps1, prob1.device), _idx(framestps2, prob1.device)
    assert f1.size(0) == f2.size(0), f'{f1.size(0)}, {f2.size(0)}'
    T = prob1.size(1)
    length = (f1[:, 1] - f1[:, 0] + 1).clamp(min=0)
    off = torch.arange(T, device=prob1.device)[None, :]
    valid = off < length[:, None]
    i1 = (f1[:, 0:1] + off).clamp(max=T - 1)
    i2 = (f2[:, 0:1] + off).clamp(max=prob2.size(1) - 1)
    p1, p2 = prob1.gather(1, i1), prob2.gather(1, i2)
    kl = torch.where(valid, p1 * torch.log((p1 + epsilon) / (p2 + epsilon)), torch.zeros_like(p1))
    return kl.sum() / f1.size(0)


def temporal_order_discrimination_loss(original_video_prob, pseudo_video_prob, criterion_domain=None):
    """2-way cross entropy, label 0 = original, 1 = shuffled (reference loss.py:6-20)."""
    o = original_video_prob.view(-1, original_video_prob.size(-1))
    p = pseudo_video_prob.view(-1, pseudo_video_prob.size(-1))
    label = torch.cat((torch.zeros(o.size(0), dtype=torch.long, device=o.device),
                       torch.ones(p.size(0), dtype=torch.long, device=o.device)))
    pred = torch.cat((o, p), 0)
    return F.cross_entropy(pred, label) if criterion_domain is None else criterion_domain(pred, label)


def span_pred(start_prob, end_prob):
    """argmax_{i<=j}(start_i + end_j): the lower triangle is zero-filled and takes part in the max,
    first maximum wins (reference loss.py:53-70).  -> (int64 [B,2], score [B]) on the input device."""
    B, T = start_prob.shape
    if start_prob.is_cuda and start_prob.dtype == torch.float32 and end_prob.dtype == torch.float32 and T <= 16384:
        # one launch, one workgroup per pair (csrc/input_pipeline.hip: tsg_span_pred) instead of a [B,T,T] matrix and six ops
        from . import _lib
        from .functional import _call
        s, e = start_prob.detach().contiguous(), end_prob.detach().contiguous()
        pred = torch.empty(B, 2, device=s.device, dtype=torch.long)
        score = torch.empty(B, device=s.device, dtype=torch.float32)
        _call("tsg_span_pred", s, s.data_ptr(), e.data_ptr(), pred.data_ptr(), score.data_ptr(), B, T, _lib.TSG_F32)
        return pred, score
    m = (start_prob.unsqueeze(2) + end_prob.unsqueeze(1)).triu(0)
    row_max, row_idx = m.max(dim=2)
    best, col = row_max.max(dim=1)
    end = row_idx.gather(1, col[:, None]).squeeze(1)
    return torch.stack((col, end), -1), best


def compute_mean_iou(seg1, seg2):
    """batch mean IoU with union = max_end - min_beg (+1e-4), reference loss.py:72-91."""
    s1, e1, s2, e2 = seg1[:, 0], seg1[:, 1], seg2[:, 0], seg2[:, 1]
    inter = (torch.minimum(e1, e2) - torch.maximum(s1, s2)).clamp(min=0)
    union = torch.maximum(e1, e2) - torch.minimum(s1, s2)
    return (inter / (union + DELTA)).mean()
