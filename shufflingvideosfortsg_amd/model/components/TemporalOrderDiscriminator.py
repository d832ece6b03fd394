"""Temporal-order discriminator (reference components/TemporalOrderDiscriminator.py), GMD only:
masked means over the target / fore / back clip ranges -> small MLPs -> 2-way logits
(original vs shuffled video).  Training-only auxiliary head: the three masked means are ONE HIP pass over the clip features
(tsg_moment_pool_fwd / _bwd, csrc/moment_pool.hip); the two small Linears on the [B, 2d] / [B, 3d] pooled rows stay torch ops."""
import torch
import torch.nn as nn

from ... import functional as TF
from ..networks.attention import mask_logits


def select_temporal_order_discriminator(name, logger):
    if name.lower() in ['moment_pooling', 'mp']:
        return MomentPooling
    logger.error('error temporal order discriminator name: %s (must be \'moment_pooling\')', name)
    raise ValueError(name)


class MomentPooling(nn.Module):
    def __init__(self, visual_dim, logger, *args):
        super().__init__()
        self.foreback_context = nn.Sequential(nn.Linear(visual_dim * 2, visual_dim), nn.ReLU(inplace=True))
        self.dropout = nn.Dropout(p=0.5)
        self.fc_classifier_domain_video = nn.Sequential(nn.Linear(visual_dim * 3, 2))

    def average_mask(self, feat, mask):
        return torch.sum(mask_logits(feat, mask, mask_value=0.0), dim=1) / (torch.sum(mask, dim=1, keepdim=True) + 1e-6)

    def average_masks(self, feat, masks):
        """The masked means of ``average_mask`` for several masks at once: mask_logits(feat, m, 0) summed over time is the
        batched product m^T feat, so the three ranges are ONE [B,3,T] x [B,T,D] bmm that reads feat once (and one bmm back),
        instead of four [B,T,D] elementwise passes and a reduction per range."""
        M = torch.stack([m.type_as(feat) for m in masks], 1)                       # [B,K,T]
        return torch.bmm(M, feat) / (M.sum(2, keepdim=True) + 1e-6)               # [B,K,D]

    def forward(self, feat, target_mask, fore_mask, back_mask):
        if (feat.is_cuda and feat.dim() == 3 and target_mask.dim() == 2 and feat.size(-1) % 4 == 0 and feat.dtype in (torch.float32, torch.bfloat16)
                and not torch.is_autocast_enabled()):
            tgt, fore_avg, back_avg = TF.moment_pool(feat, target_mask, fore_mask, back_mask)   # [B,D] fp32 each, one pass over feat
        elif feat.dim() == 3 and target_mask.dim() == 2:
            pooled = self.average_masks(feat, (target_mask, fore_mask, back_mask)).float()      # [B,3,D]: the small MLPs stay fp32
            tgt, fore_avg, back_avg = pooled[:, 0], pooled[:, 1], pooled[:, 2]
        else:
            tgt, fore_avg, back_avg = (self.average_mask(feat, m) for m in (target_mask, fore_mask, back_mask))
        fore = self.foreback_context(torch.cat((fore_avg, tgt), -1))
        back = self.foreback_context(torch.cat((tgt, back_avg), -1))
        return self.fc_classifier_domain_video(self.dropout(torch.cat((tgt, fore, back), -1)))
