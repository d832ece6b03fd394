"""Cross-modal semantic matching gate "csmm" (reference components/DistributionAlign.py), GMD only.
Adjacent to the hot path (SURVEY.md 8f "next #2"): torch ops, but with the same split-W trick as the
boundary head, so ``cat([video, sent.expand(T)])`` is not materialised unless asked for.
The selectors keep the reference's behaviour: 'cross' is always the concat, 'predict' always the MLP."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as TF
from ..networks.RNN import BiLSTM


def select_activation(name):
    return {'relu': nn.ReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid}.get(name.lower(), nn.ReLU)


class VideoTextConcat(nn.Module):
    def __init__(self, cross):
        super().__init__()
        self.output_dim = cross['video_dim'] + cross['query_dim']

    def forward(self, video_feat, query_feat):
        B, T, _ = video_feat.size()
        assert query_feat.dim() in (2, 3)
        if query_feat.dim() == 2:
            query_feat = query_feat.unsqueeze(1).expand(B, T, -1)
        elif query_feat.size(1) == 1:
            query_feat = query_feat.expand(B, T, -1)
        return torch.cat([video_feat, query_feat], dim=2)


class NoTemporal(nn.Module):
    def __init__(self, temporal):
        super().__init__()
        self.output_dim = temporal['input_dim']

    def forward(self, cross_feat):
        return cross_feat


class LSTMTemporal(nn.Module):
    def __init__(self, temporal):
        super().__init__()
        self.lstm = BiLSTM(temporal['input_dim'], temporal['hidden_dim'], temporal['layers'], temporal['dropout'])
        self.output_dim = temporal['hidden_dim'] * 2

    def forward(self, input, *args):
        return self.lstm(input)[0]


class TwoLayerdMLP(nn.Module):
    def __init__(self, predict):
        super().__init__()
        self.activation = select_activation(predict['activation'])
        self.predict = nn.Sequential(nn.Linear(predict['input_dim'], predict['hidden_dim']), self.activation(),
                                     nn.Linear(predict['hidden_dim'], 1))

    def forward(self, input, *args):
        return self.predict(input).squeeze(dim=2)


def select_cross(name):
    return VideoTextConcat


def select_temporal(name):
    return LSTMTemporal if name.lower() in ['lstm'] else NoTemporal


def select_predict(name):
    return TwoLayerdMLP


class VideoTextSemanticMatch(nn.Module):
    """-> (per-clip raw matching logits [B,T], temporal feature or None).  The reference returns the
    concatenated feature as second output and never uses it; set ``keep_temporal_feat`` to get it."""

    def __init__(self, cross, temporal, predict):
        super().__init__()
        self.cross = select_cross(cross['name'])(cross)
        temporal['input_dim'] = self.cross.output_dim
        self.temporal = select_temporal(temporal['name'])(temporal)
        predict['input_dim'] = self.temporal.output_dim
        self.predict = select_predict(predict['name'])(predict)
        self.temporal_dim = self.temporal.output_dim
        self.keep_temporal_feat = False

    def forward(self, video_feat, query_feat, video_mask):
        fast = isinstance(self.temporal, NoTemporal) and query_feat.dim() == 2 and not self.keep_temporal_feat
        if not fast:
            temporal_feat = self.temporal(self.cross(video_feat, query_feat))
            return self.predict(temporal_feat, query_feat), temporal_feat
        lin1, act, lin2 = self.predict.predict[0], self.predict.predict[1], self.predict.predict[2]
        Dv = video_feat.size(-1)
        act_name = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Sigmoid: "sigmoid"}.get(type(act))
        H = lin1.weight.size(0)
        if video_feat.is_cuda and act_name is not None and H % 4 == 0 and H <= 1024 and lin2.weight.size(0) == 1:
            B, T = video_feat.shape[:2]
            if video_feat.dtype == torch.float32 and video_feat.dim() == 3 and TF.head_gemm_ok(B * T, H, Dv, T, H):
                # K5 as the EPILOGUE of the video half's GEMM (tsg_match_head_gemm): the accumulator tile goes through add +
                # activation + the 1-output Linear in registers; W1[:, :Dv] is read in place; y exists only for the backward
                return TF.match_head_params(video_feat, query_feat, lin1.weight, lin1.bias, lin2.weight, lin2.bias, act_name), None
            cs = TF.linear(query_feat, lin1.weight[:, Dv:], lin1.bias)
            # K5: add + activation + the 1-output Linear in one pass over the video half's GEMM output (and one pass back)
            y = TF.linear(video_feat, lin1.weight[:, :Dv])
            return TF.match_head(y, cs, lin2.weight, lin2.bias, act_name), None
        hid = TF.linear(video_feat, lin1.weight[:, :Dv]) + F.linear(query_feat, lin1.weight[:, Dv:], lin1.bias).unsqueeze(1)
        # (measured and dropped: the 1-output Linear as torch.matmul(hid, w) -- its mv / ger backward is slower than the two
        # degenerate GEMMs: 17.24 vs 17.04 ms per step)
        return lin2(act(hid)).squeeze(dim=2), None
