"""Sentence side (reference components/SentenceEncoder.py): Linear(300,300) -> BiLSTM ->
(word_feat [B,N,2h], sent_embed [B,2h]).  Produces K/V for the hot path; stays on torch."""
import torch
import torch.nn as nn

from ..networks.RNN import BiLSTM


def select_sent_encoder(name, logger):
    if name.lower() in ['rnn', 'r']:
        return RNNEncoder
    logger.error('error sentence encoder name: %s (must be \'rnn\')', name)
    raise ValueError(name)


class RNNEncoder(nn.Module):
    def __init__(self, sent_seq_set, logger, *args):
        super().__init__()
        input_dim = sent_seq_set['input_dim']
        self.drop_out = sent_seq_set['drop_out']
        self.word_embed = nn.Linear(input_dim, input_dim)
        self.rnn_cell = BiLSTM(input_dim, sent_seq_set['rnn_hidden_dim'], sent_seq_set['rnn_layers'], self.drop_out)
        self.textual_dim = sent_seq_set['rnn_hidden_dim'] * 2

    def forward(self, input):
        word_encoding, final, _ = self.rnn_cell(self.word_embed(input), states="final")     # final = cat(hn[-2], hn[-1])
        return word_encoding, final
