"""Recurrent glue (reference grounding/model/networks/RNN.py:26-48).  Adjacent to the hot path:
``torch.nn.LSTM`` (MIOpen on ROCm); SURVEY.md 8f ranks a persistent HIP kernel for it as "next #1"."""
import torch.nn as nn


class BiLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, dropout=0.5):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True, dropout=dropout)

    def forward(self, x, h0=None, c0=None):
        """-> (out [B,L,2h], hn [2*layers,B,h], cn); zero initial state on x's device (the reference
        allocates it with a hard ``.cuda()``)."""
        self.lstm.flatten_parameters()
        state = None if (h0 is None or c0 is None) else (h0, c0)
        out, (hn, cn) = self.lstm(x, state)
        return out, hn, cn
