"""Query-aware video encoder (QAVE), reference components/VideoEncoder.py.

``rnn_recalibration_layer``: BiLSTM -> SCDM cross-attention (fused HIP kernel K1) -> channel gate
``rnn_out * sigmoid(sent_linear(C))``.  ``QueryAwareEncoder``: two chained blocks sharing the word
features, then LayerNorm.  Parameter names match the reference (``blocks.{i}.attention.W_s.weight`` ...).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as TF
from ..networks.RNN import BiLSTM
from ..networks.attention import SCDM_Attention


def select_video_encoder(name, logger):
    name = name.lower()
    if name in ['rnn', 'r']:
        return RNNEncoder
    if name in ['query_aware_encoder', 'qae', 'qave']:
        return QueryAwareEncoder
    logger.error('error video encoder name: %s (must be \'rnn\' or \'qae\')', name)
    raise ValueError(name)


class RNNEncoder(nn.Module):
    """Plain BiLSTM + LayerNorm alternative (no query information)."""

    def __init__(self, video_seq_set, logger, *args):
        super().__init__()
        hidden_dim = video_seq_set['rnn_hidden_dim']
        self.rnn_cell = BiLSTM(video_seq_set['input_dim'], hidden_dim, video_seq_set['rnn_layers'], video_seq_set['drop_out'])
        self.visual_dim = hidden_dim * 2
        self.video_layernorm = nn.LayerNorm(hidden_dim * 2)

    def forward(self, input, *args):
        video_encoding, _, _ = self.rnn_cell(input, states=False)
        ln = self.video_layernorm
        if TF.layer_norm_ok(video_encoding):
            return TF.layer_norm(video_encoding, ln.weight, ln.bias, ln.eps)
        if video_encoding.dtype != ln.weight.dtype:
            return F.layer_norm(video_encoding, ln.normalized_shape, ln.weight.to(video_encoding.dtype), ln.bias.to(video_encoding.dtype), ln.eps)
        return ln(video_encoding)


class rnn_recalibration_layer(nn.Module):
    def __init__(self, input_dim, sent_dim, hidden_dim, n_layers, ca_activ, drop_out, logger):
        super().__init__()
        self.ca_activ = ca_activ
        self.rnn_cell = BiLSTM(input_dim, hidden_dim, n_layers, drop_out)
        self.visual_dim = hidden_dim * 2
        self.attention = SCDM_Attention(self.visual_dim, sent_dim)
        self.sent_linear = nn.Linear(sent_dim, self.visual_dim)

    def forward(self, video_feat, word_feat):
        rnn_output, _, _ = self.rnn_cell(video_feat, states=False)
        att = self.attention
        if self.ca_activ in ['sigmoid'] and type(att) is SCDM_Attention and word_feat.size(-1) == self.sent_linear.in_features:
            # fused tail: sent_linear(P @ words) = P @ (words W_l^T) + b_l, so the Linear runs on the N word rows
            # instead of the T clip rows and bias / sigmoid / gate are the attention kernel's epilogue
            VW = TF.linear(word_feat, self.sent_linear.weight)
            if TF.scdm_gate_proj_ok(rnn_output, att.W_a.weight, VW):
                # W_a's projection inside the gate's autograd node: the two gradients of rnn_output (through W_a, and as the gate's r) are summed
                # by the input-gradient GEMM's epilogue, not by an elementwise kernel
                s = TF.linear(word_feat, att.W_s.weight, att.W_a.bias)
                return TF.scdm_gate_proj(rnn_output, att.W_a.weight, s, att.w.weight, VW, self.sent_linear.bias)
            # (other modes: the BiLSTM output is W_a's input and the gate's r -- its two gradients meet in a sink where the consumers know about
            #  sinks (bf16 storage: K1g leaves dr there, W_a's input-gradient GEMM adds onto it); elsewhere the context is a no-op)
            with TF.shared_grad(rnn_output) as xs:
                a, s = att.projections(xs, word_feat)
                return TF.scdm_gate(a, s, att.w.weight, VW, self.sent_linear.bias, xs)
        # un-fused tail (another attention class, or a word width sent_linear was not built for)
        acts = {'sigmoid': torch.sigmoid, 'relu': torch.relu, 'tanh': torch.tanh}
        channel_attn = self.sent_linear(self.attention(rnn_output, word_feat))
        return rnn_output * acts.get(self.ca_activ, lambda x: x)(channel_attn)


class QueryAwareEncoder(nn.Module):
    def __init__(self, video_seq_set, logger, *args):
        super().__init__()
        hidden_dim = video_seq_set['rnn_hidden_dim']
        self.nblocks = video_seq_set['nblocks']
        input_dim = video_seq_set['input_dim']
        self.blocks = nn.ModuleList()
        for _ in range(self.nblocks):
            self.blocks.append(rnn_recalibration_layer(input_dim, video_seq_set['query_dim'], hidden_dim,
                                                       video_seq_set['rnn_layers'], 'sigmoid',
                                                       video_seq_set['drop_out'], logger))
            input_dim = hidden_dim * 2
        self.visual_dim = hidden_dim * 2
        self.norm = nn.LayerNorm(self.visual_dim)

    def forward(self, video_feat, query_feat, *args):
        if not isinstance(query_feat, list):
            queries = [query_feat] * self.nblocks
        elif len(query_feat) < self.nblocks:
            queries = query_feat + [query_feat[-1]] * (self.nblocks - len(query_feat))
        else:
            queries = query_feat
        x = video_feat
        for blk, q in zip(self.blocks, queries):
            x = blk(x, q)
        if TF.layer_norm_ok(x):                              # one pass forward, one pass backward (csrc/layer_norm.hip); bf16 storage: bf16 in / out
            return TF.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        if x.dtype != self.norm.weight.dtype:                # bf16 storage mode: bf16 in and out, fp32 statistics inside the kernel
            return F.layer_norm(x, self.norm.normalized_shape, self.norm.weight.to(x.dtype), self.norm.bias.to(x.dtype), self.norm.eps)
        return self.norm(x)
