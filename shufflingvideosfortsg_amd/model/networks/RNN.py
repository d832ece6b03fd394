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


class BiLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout=0.5):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True, dropout=dropout)
        # "hip" (default on GPU tensors) | "miopen" = torch.nn.LSTM, kept for A/B timing
        self.backend = os.environ.get("TSG_LSTM", "hip")

    def _hip_forward(self, x):
        L, p = self.lstm, self.lstm.dropout
        bm = os.environ.get("TSG_LSTM_LAYOUT", "bm") != "tm"   # default: batch-major throughout, the kernels index [B,T,..]
        inp, hn, cn = (x if bm else x.transpose(0, 1).contiguous()), [], []     # directly; "tm" = transposed copies (A/B timing)
        for k in range(self.num_layers):
            g = lambda n: getattr(L, f"{n}_l{k}")
            gr = lambda n: getattr(L, f"{n}_l{k}_reverse")
            W_ih = torch.cat([g("weight_ih"), gr("weight_ih")], 0)
            bias = torch.cat([g("bias_ih") + g("bias_hh"), gr("bias_ih") + gr("bias_hh")])
            W_hh = torch.stack([g("weight_hh"), gr("weight_hh")])
            out, Cs = TF.bilstm_layer(inp, W_ih, bias, W_hh, batch_major=bm)
            h = self.hidden_size
            hn += [out[:, -1, :h], out[:, 0, h:]] if bm else [out[-1, :, :h], out[0, :, h:]]
            cn += [Cs[-1, 0], Cs[0, 1]]
            inp = F.dropout(out, p, self.training) if (p > 0 and k + 1 < self.num_layers) else out
        return (out if bm else out.transpose(0, 1).contiguous()), torch.stack(hn, 0), torch.stack(cn, 0)

    def forward(self, x, h0=None, c0=None):
        """-> (out [B,L,2h], hn [2*layers,B,h], cn); zero initial state on x's device (the reference
        allocates it with a hard ``.cuda()``)."""
        if x.is_cuda and self.backend == "hip" and h0 is None and c0 is None:
            return self._hip_forward(x)
        self.lstm.flatten_parameters()
        state = None if (h0 is None or c0 is None) else (h0, c0)
        out, (hn, cn) = self.lstm(x, state)
        return out, hn, cn
