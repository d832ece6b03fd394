"""Boundary (start/end) predictors, reference components/SpanPredictor.py.

In scope: ``SpanPredictor_Boundary`` (dispatcher), ``MLP_predictor`` (the live head; fused kernel K3)
and ``Self_Attention_predictor`` (temporal self-attention head on kernel K2 -- unconstructible in the
reference; here it works and takes the ``v_mask`` its caller passes).  The LSTM-based
alternates ('tied_lstm', 'condi_lstm', ...) are out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as TF
from ..networks.attention import MultiHead, mask_logits, positional_encodings_like


class SpanPredictor_Boundary(nn.Module):
    def __init__(self, crossmodal_dim, predictor_set, drop_out, logger):
        super().__init__()
        self.crossmodal_dim = crossmodal_dim
        self.drop_out = drop_out
        name = predictor_set['name']
        if name in ['mlp', 'a']:
            self.predictor = MLP_predictor(crossmodal_dim, predictor_set['mlp_hidden_dim'])
        elif name in ['self_attn', 'd']:
            self.predictor = Self_Attention_predictor(crossmodal_dim, predictor_set['attention_nheads'],
                                                      predictor_set['position_encoding'], drop_out)
        else:
            if logger is not None:
                logger.error('predictor %r is not provided by this build (supported: mlp, self_attn)', name)
            raise NotImplementedError(f"span predictor {name!r}: only 'mlp'/'a' and 'self_attn'/'d' are implemented")

    def forward(self, crossmodal_feat, v_mask=None):
        return self.predictor(crossmodal_feat, v_mask)

    def forward_split(self, video_feat, sent_feat, gate=None, v_mask=None):
        """Same result as ``forward(gate * cat(video, sent), v_mask)`` without building the concat."""
        if isinstance(self.predictor, MLP_predictor):
            return self.predictor.forward_split(video_feat, sent_feat, gate, v_mask)
        x = torch.cat([video_feat, sent_feat.unsqueeze(1).expand(-1, video_feat.size(1), -1)], dim=-1)
        if gate is not None:
            x = gate.unsqueeze(2) * x
        return self.predictor(x, v_mask)


class MLP_predictor(nn.Module):
    """Two heads ``Linear(D,Hm) -> tanh -> Linear(Hm,1)``, optional mask_logits, softmax over T
    (reference SpanPredictor.py:60-85).  Everything after the first GEMM is one fused kernel."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.start_mlp_1 = nn.Linear(input_dim, hidden_dim)
        self.start_mlp_2 = nn.Linear(hidden_dim, 1)
        self.end_mlp_1 = nn.Linear(input_dim, hidden_dim)
        self.end_mlp_2 = nn.Linear(hidden_dim, 1)

    def _stacked(self):
        W1 = torch.cat([self.start_mlp_1.weight, self.end_mlp_1.weight], 0)          # [2Hm, D]
        b1 = torch.cat([self.start_mlp_1.bias, self.end_mlp_1.bias])
        w2 = torch.cat([self.start_mlp_2.weight.reshape(-1), self.end_mlp_2.weight.reshape(-1)])
        b2 = torch.cat([self.start_mlp_2.bias, self.end_mlp_2.bias])
        return W1, b1, w2, b2

    def forward(self, crossmodal_feat, v_mask=None):
        W1, b1, w2, b2 = self._stacked()
        y = TF.linear(crossmodal_feat, W1)
        cs = y.new_zeros(y.size(0), y.size(2))
        return TF.boundary_score(y, cs, b1, w2, b2, None, v_mask)

    def forward_split(self, video_feat, sent_feat, gate=None, v_mask=None):
        """video [B,T,Dv], sent [B,Ds] (Dv+Ds = input_dim), gate [B,T] or None: the sentence half of
        the first Linear is a per-sample row, the [B,T,Dv+Ds] concat is never built."""
        Dv = video_feat.size(-1)
        Hm = self.hidden_dim
        if (video_feat.is_cuda and video_feat.dtype == torch.float32 and video_feat.dim() == 3
                and TF.head_gemm_ok(video_feat.size(0) * video_feat.size(1), 2 * Hm, Dv, video_feat.size(1), Hm)):
            # K3 as the EPILOGUE of the video half's GEMM (tsg_boundary_head_gemm): the two heads' first Linears are read in place
            # (no stacked copy, no column-slice nodes); only the [B,T] probabilities leave the kernel
            return TF.boundary_head_params(video_feat, sent_feat, self.start_mlp_1.weight, self.start_mlp_1.bias, self.end_mlp_1.weight,
                                    self.end_mlp_1.bias, self.start_mlp_2.weight, self.start_mlp_2.bias, self.end_mlp_2.weight,
                                    self.end_mlp_2.bias, gate, v_mask)
        W1, b1, w2, b2 = self._stacked()
        y = TF.linear(video_feat, W1[:, :Dv])
        cs = TF.linear(sent_feat, W1[:, Dv:])
        return TF.boundary_score(y, cs, b1, w2, b2, gate, v_mask)


class Self_Attention_predictor(nn.Module):
    """Temporal self-attention over the fused clip features -> boundary scores (reference
    SpanPredictor.py:244-266): optional sinusoid position table, two ``MultiHead(x,x,x)``,
    ``Linear(D,1)``, softmax over T.  ``v_mask`` (ignored by the reference's signature, which its
    caller nevertheless passes) is applied with ``mask_logits`` when given."""

    def __init__(self, input_dim, n_heads, position_encoding, drop_out):
        super().__init__()
        self.crossmodal_dim = input_dim
        self.position_encoding = position_encoding
        self.start_selfattn = MultiHead(input_dim, input_dim, n_heads, drop_out)
        self.end_selfattn = MultiHead(input_dim, input_dim, n_heads, drop_out)
        self.start_fc = nn.Linear(input_dim, 1)
        self.end_fc = nn.Linear(input_dim, 1)

    def forward(self, crossmodal_feat, v_mask=None):
        x = crossmodal_feat
        if self.position_encoding:
            x = x + positional_encodings_like(x)
        # (bf16 storage mode: the attention output is bf16; the [D -> 1] logits, their mask and the softmax over T are fp32 like every output the losses read)
        s = self.start_fc(self.start_selfattn(x, x, x).to(self.start_fc.weight.dtype)).squeeze(dim=2)
        e = self.end_fc(self.end_selfattn(x, x, x).to(self.end_fc.weight.dtype)).squeeze(dim=2)
        if v_mask is not None:
            s, e = mask_logits(s, v_mask), mask_logits(e, v_mask)
        return torch.softmax(s, dim=1), torch.softmax(e, dim=1)
