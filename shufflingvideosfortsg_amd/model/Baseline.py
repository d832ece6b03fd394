"""QAVE baseline grounding model (reference grounding/model/Baseline.py): sentence encoder ->
query-aware video encoder (K1) -> boundary head (K3).  Same constructor protocol (the four setting
dicts of train_baseline.py:50-91), parameter names and ``forward`` / ``eval_forward`` signatures.
``query_mask`` / ``sent_mask`` are accepted and ignored, as in the reference."""
import torch.nn as nn

from .components import CrossModalInteraction, SentenceEncoder, SpanPredictor, VideoEncoder


class Baseline(nn.Module):
    def __init__(self, video_seq_set, sent_seq_set, grounding_set, matching_set, logger, drop_out):
        super().__init__()
        self.sentence_encoder = SentenceEncoder.select_sent_encoder(sent_seq_set['name'], logger)(sent_seq_set, logger)
        self.textual_dim = self.sentence_encoder.textual_dim

        video_seq_set['query_dim'] = self.textual_dim
        self.query_level_in_video = 'word'
        self.video_encoder = VideoEncoder.select_video_encoder(video_seq_set['name'], logger)(video_seq_set, logger)
        self.visual_dim = self.video_encoder.visual_dim
        self.video_if_mask = video_seq_set['mask']

        self.CMI = CrossModalInteraction.select_CMI(grounding_set['cross_name'], logger)(self.visual_dim, self.textual_dim)
        self.cross_dim = self.CMI.cross_dim()
        self.span_predictor = SpanPredictor.SpanPredictor_Boundary(self.cross_dim, grounding_set, drop_out=drop_out, logger=logger)
        self._fused_head = isinstance(self.CMI, CrossModalInteraction.VideoSentenceConcat)

    def forward(self, video_feat, query_feat, video_mask=None, query_mask=None):
        word_feature, sent_embed = self.sentence_encoder(query_feat)
        frame_feature = self.video_encoder(video_feat, word_feature)
        mask = video_mask if self.video_if_mask else None
        if self._fused_head:        # [video_t | sent] is consumed without being materialised
            start_prob, end_prob = self.span_predictor.forward_split(frame_feature, sent_embed, None, mask)
        else:
            start_prob, end_prob = self.span_predictor(self.CMI(frame_feature, word_feature, sent_embed), v_mask=mask)
        return {'start': start_prob, 'end': end_prob}

    def eval_forward(self, video_feat, sent_feat, video_mask=None, sent_mask=None):
        return self.forward(video_feat, sent_feat, video_mask, sent_mask)
