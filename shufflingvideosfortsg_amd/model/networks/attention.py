"""Attention modules of the cross-modal matching path, MI355X-native.

Same names / constructor arguments / parameter names / forward signatures as the reference's
``grounding/model/networks/attention.py`` so its ``state_dict``s load unchanged; the arithmetic
between the projections runs in hand-written HIP kernels (libtsg_hip.so, include/tsg_hip.h).
There is no CPU path: every forward needs device tensors.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as TF

INF = 1e10


def positional_encodings_like(x: torch.Tensor, t: torch.Tensor = None) -> torch.Tensor:
    """[T, D] sinusoid table for ``x`` = [B, T, D] (reference attention.py:16-35): channel c even ->
    sin(pos / 10000^(c/D)), odd -> cos(pos / 10000^((c-1)/D)).  Built in one shot on x's device
    (the reference fills it channel by channel in a Python loop); same float32 operand types."""
    T, D = x.size(1), x.size(2)
    key = (T, D, x.device)
    if t is None and key in _PE_CACHE:                     # the table depends on (T, D) only: built once per shape and device (a host
        return _PE_CACHE[key]                              # list -> device copy on every call is also illegal under graph capture)
    pos = torch.arange(0, T, device=x.device).float() if t is None else t
    c = torch.arange(D)
    div = torch.tensor([10000 ** (float(ci - (ci % 2)) / D) for ci in c.tolist()], dtype=torch.float32, device=x.device)
    ang = pos[:, None] / div[None, :]
    even = (c % 2 == 0).to(x.device)
    pe = torch.where(even[None, :], torch.sin(ang), torch.cos(ang))
    if t is None:
        _PE_CACHE[key] = pe
    return pe


_PE_CACHE = {}


def masked_softmax(vec, mask, dim=1, epsilon=1e-4):
    """exp(x)*m / (sum exp(x)*m + eps), no max-subtraction (reference attention.py:123-127)."""
    e = torch.exp(vec) * mask.float()
    return e / (e.sum(dim, keepdim=True) + epsilon)


def mask_logits(inputs, mask, mask_value=-1e30):
    """x*m + v*(1-m) (reference attention.py:129-133)."""
    m = mask.type_as(inputs)
    if m.dim() == inputs.dim() - 1:
        m = m.unsqueeze(-1).expand(-1, -1, inputs.size(-1))
    return inputs * m + mask_value * (1.0 - m)


class Attention(nn.Module):
    """Single-head scaled dot-product attention, ``forward(q,k,v) -> (out, A, A_softmax)``
    (reference attention.py:37-55; constructible here, unlike the reference -- SURVEY.md F1).
    ``scale = sqrt(d_key)`` of whatever width the constructor is given."""

    def __init__(self, d_key, drop_ratio, causal):
        super().__init__()
        self.scale = math.sqrt(d_key)
        self.dropout = nn.Dropout(drop_ratio)
        self.causal = causal

    def _p(self):
        """Dropout probability in effect (training mode only); applied inside the kernel to the softmax."""
        return float(self.dropout.p) if self.training else 0.0

    def forward(self, query, key, value):
        if query.dim() != 3:
            raise ValueError("Attention expects [B, T, d] tensors")
        return TF.mha(query, key, value, 1, self.scale, bool(self.causal), return_maps=True, p_drop=self._p())


class MultiHead(nn.Module):
    """Multi-head attention with the reference's quirks (attention.py:57-97): bias-free wq/wk/wv/wo,
    heads = ``chunk(n_heads, -1)``, every head scaled by sqrt(d_key) of the FULL width (F2).
    All heads run in one fused kernel launch."""

    def __init__(self, d_key, d_value, n_heads, drop_ratio, causal=False):
        super().__init__()
        self.attention = Attention(d_key, drop_ratio, causal=causal)
        self.wq = nn.Linear(d_key, d_key, bias=False)
        self.wk = nn.Linear(d_key, d_key, bias=False)
        self.wv = nn.Linear(d_value, d_value, bias=False)
        self.wo = nn.Linear(d_value, d_key, bias=False)
        self.n_heads = n_heads
        self.A = None
        self.A_softmax = None

    def _core(self, query, key, value, maps):
        # TF.linear = F.linear, or the split-precision GEMMs in the f32s mode when the operand has >= 2048 rows
        q, k, v = TF.linear(query, self.wq.weight), TF.linear(key, self.wk.weight), TF.linear(value, self.wv.weight)
        return TF.mha(q, k, v, self.n_heads, self.attention.scale, bool(self.attention.causal), return_maps=maps,
                      p_drop=self.attention._p())

    def forward(self, query, key, value):
        return TF.linear(self._core(query, key, value, False), self.wo.weight)

    def A_forward(self, query, key, value):
        o, self.A, self.A_softmax = self._core(query, key, value, True)
        return TF.linear(o, self.wo.weight)


class SCDM_Attention(nn.Module):
    """Additive video<->word attention (reference attention.py:99-121):
    P[b,t,n] = w . tanh(W_s s_n + W_a v_t + b), softmax over the N words (no word mask), C = P @ sent.
    The [B,T,N,H] tanh tensor is never materialised (fused kernel K1, forward and backward)."""

    def __init__(self, video_dim, sent_dim, hidden_dim=None):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = video_dim
        self.W_s = nn.Linear(sent_dim, hidden_dim, bias=False)
        self.W_a = nn.Linear(video_dim, hidden_dim)
        self.w = nn.Linear(hidden_dim, 1, bias=False)

    def forward(self, video_feat, sent_feat):
        a, s = self.projections(video_feat, sent_feat)
        return TF.scdm_attn(a, s, self.w.weight, sent_feat)

    def projections(self, video_feat, sent_feat):
        """(W_a v, W_s s + b): the bias of W_a rides on the N word rows instead of the T clip rows -- tanh(W_s s_n + W_a v_t + b)
        is the same sum, and the [B,T,H] bias pass (and its [B*T,H] -> [H] gradient reduction) becomes a [B,N,H] one."""
        a = TF.linear(video_feat, self.W_a.weight, None)
        s = TF.linear(sent_feat, self.W_s.weight, self.W_a.bias)     # (the bias in the Linear: its GEMM's epilogue / one in-place add)
        return a, s
