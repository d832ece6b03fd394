"""Cross-modal fusion selectors (reference components/CrossModalInteraction.py).  The live choice
'vs' (``VideoSentenceConcat``) builds ``[video_t | sent]``; the models do not call it on the fast
path -- the boundary head consumes (video, sent) separately (kernel K3) -- but the module still
returns the concatenated tensor when used standalone."""
import torch
import torch.nn as nn


def select_CMI(name, logger):
    key = name.lower()
    if key in ['onlyvideo', 'a']:
        return OnlyVideo
    if key in ['videosentconcat', 'vs', 'b']:
        return VideoSentenceConcat
    if key in ['tall', 'mm', 'c']:
        return TALL
    logger.error('error CMI name: %s (must be a, b or c)', name)
    raise ValueError(name)


class _CMI(nn.Module):
    def __init__(self, video_dim, sent_dim, cross_dim):
        super().__init__()
        self.video_dim, self.sent_dim, self._cross_dim = video_dim, sent_dim, cross_dim

    def cross_dim(self):
        return self._cross_dim


class OnlyVideo(_CMI):
    def __init__(self, video_dim, sent_dim, *args):
        super().__init__(video_dim, sent_dim, video_dim)

    def forward(self, video_feat, word_feat, sent_feat):
        return video_feat


class VideoSentenceConcat(_CMI):
    def __init__(self, video_dim, sent_dim, *args):
        super().__init__(video_dim, sent_dim, video_dim + sent_dim)

    def forward(self, video_feat, word_feat, sent_feat):
        return torch.cat([video_feat, sent_feat.unsqueeze(1).expand(-1, video_feat.size(1), -1)], dim=-1)


class TALL(_CMI):
    def __init__(self, video_dim, sent_dim, *args):
        assert video_dim == sent_dim
        super().__init__(video_dim, sent_dim, video_dim * 4)
        self.crossmodal_dim = self._cross_dim

    def forward(self, video_feat, word_feat, sent_feat):
        s = sent_feat.unsqueeze(1).expand(-1, video_feat.size(1), -1)
        return torch.cat((video_feat, s, video_feat * s, video_feat + s), -1)
