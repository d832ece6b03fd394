"""GMD: QAVE + cross-modal matching gate + temporal-order discriminator, trained on (original video,
shuffled pseudo video) pairs (reference grounding/model/SpanGroundMatchDisc.py).  Same constructor
protocol, parameter names and ``forward`` (12 tensors -> 5-tuple) / ``eval_forward`` signatures.
The raw matching logits gate the fused feature (SpanGroundMatchDisc.py:86); here the gate goes
straight into the boundary kernel (K3), the gated concat is never built."""
import torch
import torch.nn as nn

from .. import functional as TF
from ..data import adjacent_cat
from .components import (CrossModalInteraction, SentenceEncoder, SpanPredictor, TemporalOrderDiscriminator,
                         VideoEncoder)
from .components.DistributionAlign import VideoTextSemanticMatch


class GMD(nn.Module):
    def __init__(self, video_seq_set, sent_seq_set, grounding_set, matching_set, logger, drop_out):
        super().__init__()
        self.sentence_encoder = SentenceEncoder.select_sent_encoder(sent_seq_set['name'], logger)(sent_seq_set, logger)
        self.textual_dim = self.sentence_encoder.textual_dim

        video_seq_set['query_dim'] = self.textual_dim
        self.video_encoder = VideoEncoder.select_video_encoder(video_seq_set['name'], logger)(video_seq_set, logger)
        self.visual_dim = self.video_encoder.visual_dim
        self.video_if_mask = video_seq_set['mask']

        self.CMI = CrossModalInteraction.select_CMI(grounding_set['cross_name'], logger)(self.visual_dim, self.textual_dim)
        self.cross_dim = self.CMI.cross_dim()
        self.span_predictor = SpanPredictor.SpanPredictor_Boundary(self.cross_dim, grounding_set, drop_out=drop_out, logger=logger)

        matching_set['cross']['video_dim'] = self.visual_dim
        matching_set['cross']['query_dim'] = self.textual_dim
        self.csmm = VideoTextSemanticMatch(matching_set['cross'], matching_set['temporal'], matching_set['predict'])
        self.matching_dim = self.csmm.temporal_dim

        tod = TemporalOrderDiscriminator.select_temporal_order_discriminator('moment_pooling', logger)
        self.tod = tod(self.visual_dim, logger)
        self._fused_head = isinstance(self.CMI, CrossModalInteraction.VideoSentenceConcat)

    def _span(self, frame_feat, word_feat, sent_embed, gate, video_mask):
        mask = video_mask if self.video_if_mask else None
        if self._fused_head:
            s, e = self.span_predictor.forward_split(frame_feat, sent_embed, gate, mask)
        else:
            s, e = self.span_predictor(gate.unsqueeze(dim=2) * self.CMI(frame_feat, word_feat, sent_embed), v_mask=mask)
        return {'start': s, 'end': e}

    def forward(self, query_feat, query_mask, ori_video_feat, ori_video_mask, pseudo_video_feat, pseudo_video_mask,
                ori_temporal_mask, ori_fore_mask, ori_back_mask, pseudo_temporal_mask, pseudo_fore_mask, pseudo_back_mask):
        word_feat, sent_embed = self.sentence_encoder(query_feat)
        # the original and the shuffled video go through the shared-weight encoder as ONE batch of 2B
        # (every op in it is per-sample): half as many sequential LSTM steps, twice the rows per launch
        B = ori_video_feat.size(0)
        both = self.video_encoder(adjacent_cat(ori_video_feat, pseudo_video_feat), torch.cat([word_feat, word_feat], 0))
        # the clip features feed three consumers (matching head, boundary head on the first B rows, temporal-order discriminator): their input
        # gradients are summed inside their own kernels (TF.shared_grad / GradSink), not by autograd's add kernels
        with TF.shared_grad(both) as both:
            ori_frame_feat, pseudo_frame_feat = both[:B], both[B:]
            # the matching gate and the temporal-order discriminator are per-sample too: both streams in one pass
            cat2 = lambda a, b: None if a is None or b is None else adjacent_cat(a, b)     # (a view when the batch put them back to back)
            vmask2 = cat2(ori_video_mask, pseudo_video_mask)
            if vmask2 is not None or (ori_video_mask is None and pseudo_video_mask is None):
                match2, _ = self.csmm(both, torch.cat([sent_embed, sent_embed], 0), vmask2)
                ori_match, pseudo_match = match2[:B], match2[B:]
            else:
                ori_match, _ = self.csmm(ori_frame_feat, sent_embed, ori_video_mask)
                pseudo_match, _ = self.csmm(pseudo_frame_feat, sent_embed, pseudo_video_mask)
            span_prob = self._span(ori_frame_feat, word_feat, sent_embed, ori_match, ori_video_mask)
            disc2 = self.tod(both, adjacent_cat(ori_temporal_mask, pseudo_temporal_mask),
                             adjacent_cat(ori_fore_mask, pseudo_fore_mask), adjacent_cat(ori_back_mask, pseudo_back_mask))
        return span_prob, ori_match, pseudo_match, disc2[:B], disc2[B:]

    def eval_forward(self, video_feat, query_feat, video_mask=None, sent_mask=None):
        word_feat, sent_embed = self.sentence_encoder(query_feat)
        frame_feat = self.video_encoder(video_feat, word_feat)
        match, _ = self.csmm(frame_feat, sent_embed, video_mask)
        return self._span(frame_feat, word_feat, sent_embed, match, video_mask)
