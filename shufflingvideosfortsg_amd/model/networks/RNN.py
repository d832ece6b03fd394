"""Recurrent glue (reference grounding/model/networks/RNN.py:26-48).  Adjacent to the hot path, but
~85 % of a training step's GPU time when left to MIOpen's per-timestep kernels, so on the GPU the
recurrence runs in libtsg_hip.so (csrc/lstm.hip: one fused launch per time step for both directions,
fp32 MFMA) with the input / weight-gradient GEMMs on rocBLAS.  Parameters are an ``nn.LSTM``'s, so
``state_dict`` keys (``lstm.weight_ih_l0`` ...) and default init equal the reference's."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as TF


class _Joined(torch.autograd.Function):
    """Two parameters that are ADJACENT views of one buffer -> the buffer viewed as their concatenation along dim 0 (``stack``:
    as a new leading axis), without a copy; the gradient goes back as the two halves (views of it).  Replaces the
    ``torch.cat`` / ``torch.stack`` of the forward and reverse ``nn.LSTM`` weights every layer call (16 MB of copies and
    five launches per layer and step)."""

    @staticmethod
    def forward(ctx, pf, pr, base, stack):
        ctx.n0 = pf.shape[0]
        ctx.stack = stack
        shape = (2,) + tuple(pf.shape) if stack else (2 * pf.shape[0],) + tuple(pf.shape[1:])
        return base.detach().view(shape)

    @staticmethod
    def backward(ctx, g):
        if ctx.stack:
            return g[0], g[1], None, None
        return g[:ctx.n0], g[ctx.n0:], None, None


class _FinalStates(torch.autograd.Function):
    """out [B,L,2h] (batch-major) -> [B,2h] = (out[:, -1, :h] | out[:, 0, h:]): the last layer's final hidden states of the two
    directions, i.e. cat(hn[-2], hn[-1]).  One cat forward; backward = one zero fill and two slice copies (as four selects of a
    stacked h_n it was four zero-filled [B,L,2h] gradients, four copies and four accumulations per step)."""

    @staticmethod
    def forward(ctx, out, h):
        ctx.shape, ctx.h = out.shape, h
        return torch.cat((out[:, -1, :h], out[:, 0, h:]), -1)

    @staticmethod
    def backward(ctx, g):
        h = ctx.h
        d = g.new_zeros(ctx.shape)
        d[:, -1, :h] = g[:, :h]
        d[:, 0, h:] = g[:, h:]
        return d, None


class BiLSTM(nn.Module):
    _PARTS = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.5):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True, dropout=dropout)
        # "hip" (default on GPU tensors) | "miopen" = torch.nn.LSTM, kept for A/B timing
        self.backend = os.environ.get("TSG_LSTM", "hip")
        self._join = os.environ.get("TSG_LSTM_JOIN", "1") != "0"      # A/B switch: 0 = torch.cat / torch.stack every call
        self._fused = {}                          # (layer, part) -> buffer holding the forward and the reverse parameter back to back
        self._shadow = {}                         # layer -> bf16 buffer holding the two weight_ih shadows back to back (bf16 storage mode)

    def _joined(self, k, part, stack=False):
        """[forward; reverse] parameter ``part`` of layer k as ONE tensor.  The two nn.Parameters keep their identity, names and
        shapes (state_dict, optimizer, gradient exchange see nothing); their ``.data`` are re-pointed once at the two halves of
        a shared buffer, and re-pointed again whenever something replaced them (``.to()``, ``flatten_parameters()``, a deep
        copy): checked by address on every call."""
        L = self.lstm
        pf, pr = getattr(L, f"{part}_l{k}"), getattr(L, f"{part}_l{k}_reverse")
        if not self._join:
            return (torch.stack if stack else torch.cat)([pf, pr], 0)
        base = self._fused.get((k, part))
        n, es = pf.numel(), pf.element_size()
        if (base is None or base.device != pf.device or base.dtype != pf.dtype or pf.data_ptr() != base.data_ptr()
                or pr.data_ptr() != base.data_ptr() + n * es or not pf.is_contiguous() or not pr.is_contiguous()):
            if torch.cuda.is_available() and pf.is_cuda and torch.cuda.is_current_stream_capturing():
                return (torch.stack if stack else torch.cat)([pf, pr], 0)      # never re-point under a graph capture
            with torch.no_grad():
                base = torch.cat([pf.detach().reshape(-1), pr.detach().reshape(-1)])
                pf.data = base[:n].view(pf.shape)
                pr.data = base[n:].view(pr.shape)
            self._fused[(k, part)] = base
        return _Joined.apply(pf, pr, base, stack)

    def _joined_bf16(self, k):
        """bf16 storage mode: the joined [forward; reverse] ``weight_ih`` of layer k as bf16 WITHOUT a cast -- one bf16 buffer whose halves are the two
        parameters' shadows (functional.weight_bf16 / shadow_of), which the optimizer's kernel rewrites with every update
        (tsg_adam_step_shadow).  None when it cannot be kept (a graph capture in progress while it would have to be (re)made)."""
        L = self.lstm
        pf, pr = getattr(L, f"weight_ih_l{k}"), getattr(L, f"weight_ih_l{k}_reverse")
        if not (TF._SHADOWS and pf.is_cuda and pf.requires_grad and pr.requires_grad and pf.dtype == torch.float32):
            return None
        n = pf.numel()
        sb = self._shadow.get(k)
        sf, sr = TF.shadow_of(pf), TF.shadow_of(pr)
        if sb is None or sf is None or sr is None or sf.data_ptr() != sb.data_ptr() or sr.data_ptr() != sb.data_ptr() + 2 * n or sb.device != pf.device:
            if torch.cuda.is_current_stream_capturing():
                return None
            with torch.no_grad():
                sb = torch.cat([pf.detach().reshape(-1), pr.detach().reshape(-1)]).to(torch.bfloat16)
            pf._tsg_shadow, pf._tsg_shadow_version = sb[:n].view(pf.shape), (pf._version, pf.data_ptr())
            pr._tsg_shadow, pr._tsg_shadow_version = sb[n:].view(pr.shape), (pr._version, pr.data_ptr())
            self._shadow[k] = sb
        return sb.view(2 * pf.shape[0], *pf.shape[1:])

    def _hip_forward(self, x, states=True):
        L, p = self.lstm, self.lstm.dropout
        bm = os.environ.get("TSG_LSTM_LAYOUT", "bm") != "tm"   # default: batch-major throughout, the kernels index [B,T,..]
        inp, hn, cn = (x if bm else x.transpose(0, 1).contiguous()), [], []     # directly; "tm" = transposed copies (A/B timing)
        for k in range(self.num_layers):
            W_ih = self._joined(k, "weight_ih")                              # [8h, I]   (forward rows, then reverse)
            W_hh = self._joined(k, "weight_hh", stack=True)                  # [2, 4h, h]
            wb16 = self._joined_bf16(k) if TF.bf16_storage() and inp.is_cuda else None
            out, Cs = TF.bilstm_layer(inp, W_ih, self._joined(k, "bias_ih"), W_hh, batch_major=bm, bias2=self._joined(k, "bias_hh"), W_ih_bf16=wb16)
            h = self.hidden_size
            if states is True:
                hn += [out[:, -1, :h], out[:, 0, h:]] if bm else [out[-1, :, :h], out[0, :, h:]]
                cn += [Cs[-1, 0], Cs[0, 1]]
            if p > 0 and k + 1 < self.num_layers:         # nn.LSTM's dropout between the layers: no stored mask (tsg_dropout)
                inp = TF.dropout(out, p, self.training) if TF.dropout_ok(out) else F.dropout(out, p, self.training)
            else:
                inp = out
        out = out if bm else out.transpose(0, 1).contiguous()
        if states is True:
            return out, torch.stack(hn, 0), torch.stack(cn, 0)
        if states == "final":                      # [B, 2h] = the last layer's (forward h_T | reverse h_1): all the sentence encoder uses
            return out, _FinalStates.apply(out, self.hidden_size), None
        return out, None, None                     # the video encoders discard h_n / c_n: no selects, no stacks, nothing to back-propagate

    def forward(self, x, h0=None, c0=None, states=True):
        """-> (out [B,L,2h], hn [2*layers,B,h], cn); zero initial state on x's device (the reference
        allocates it with a hard ``.cuda()``).  ``states`` (an addition to the reference's signature, default = the reference's
        result): False -> (out, None, None) for callers that discard the final states; "final" -> (out, [B,2h] = cat(hn[-2], hn[-1]),
        None), the one combination SentenceEncoder.RNNEncoder uses."""
        if x.is_cuda and self.backend == "hip" and h0 is None and c0 is None:
            return self._hip_forward(x, states)
        self.lstm.flatten_parameters()
        state = None if (h0 is None or c0 is None) else (h0, c0)
        out, (hn, cn) = self.lstm(x, state)
        if states == "final":
            return out, torch.cat((hn[-2], hn[-1]), -1), None
        return (out, hn, cn) if states is True else (out, None, None)
