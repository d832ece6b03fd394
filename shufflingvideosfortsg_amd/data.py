"""Input-side helpers of the grounding path: label masks, the shuffling augmentation and the
synthetic batches used by the benchmarks.  CPU/numpy like the reference's dataset
code (grounding/dataset/charades.py, data_augment.py); nothing here touches the hot-path kernels."""
from __future__ import annotations

import numpy as np
import torch


def Sequence_mask(max_len, temporal_boundary):
    """int32 mask with ones on [st, et] INCLUSIVE, clipped to the sequence (charades.py:12-18)."""
    st, et = temporal_boundary
    mask = np.zeros([max_len], dtype=np.int32)
    mask[max(0, st):min(et, max_len - 1) + 1] = 1
    return mask


def gt_moment_translate(framestps, nfeats, video_feat, cropin_start=None, rng=None):
    """The shuffling augmentation (data_augment.py:135-156): cut the ground-truth moment [s, e] out of
    the first ``nfeats`` clips, close the gap and re-insert it in front of position ``cropin_start``
    (uniform in [0, nfeats-len] when not given; ``rng`` = ``random.Random`` / ``np.random.RandomState``)
    of the gap-closed sequence.  ``video_feat`` is [1, T, D]; returns (new [s,e], nfeats, new feats).
    No-op for moments of length <= 1 or covering every clip."""
    s, e = framestps
    n = e - s + 1
    if n <= 1 or n >= nfeats:
        return list(framestps), nfeats, video_feat
    if cropin_start is None:
        if rng is None:
            import random as rng       # the reference draws from the unseeded global ``random``
        cropin_start = rng.randint(0, nfeats - n)
    order = np.concatenate([np.arange(0, s), np.arange(e + 1, nfeats)])           # gap closed
    order = np.concatenate([order[:cropin_start], np.arange(s, e + 1), order[cropin_start:]])
    out = np.zeros(video_feat.shape) + 0.0
    out[0, :nfeats] = video_feat[0, order]
    return [cropin_start, cropin_start + n - 1], nfeats, out


def synthetic_batch(B, T, N, video_dim=1024, word_dim=300, seed=1234, pair=False, device="cpu"):
    """Seeded synthetic (video, query, masks, labels) batch with the shapes and label conventions of
    the reference's collate functions; ``pair=True`` adds the shuffled pseudo video of the GMD step."""
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    video = torch.randn(B, T, video_dim, generator=g)
    query = torch.randn(B, N, word_dim, generator=g) * 0.4
    nfeats = rs.randint(max(2, T // 2), T + 1, size=B)
    fs = []
    for n in nfeats:
        s = rs.randint(0, n - 1); e = rs.randint(s + 1, n)
        fs.append([int(s), int(e)])

    def labels(fsl):
        return {"framestps": [list(f) for f in fsl],
                "temporal_labels": torch.from_numpy(np.stack([Sequence_mask(T, f) for f in fsl])),
                "fore_masks": torch.from_numpy(np.stack([Sequence_mask(T, [0, f[0]]) for f in fsl])),
                "back_masks": torch.from_numpy(np.stack([Sequence_mask(T, [f[1], int(n)]) for f, n in zip(fsl, nfeats)]))}
    out = {"video": video, "query": query, "nfeats": nfeats,
           "video_mask": torch.from_numpy(np.stack([Sequence_mask(T, [0, int(n)]) for n in nfeats])),
           "query_mask": torch.ones(B, N, dtype=torch.int32), "gt": labels(fs)}
    out["gt"]["timestps"] = torch.tensor(fs, dtype=torch.float32)
    if pair:
        pv, pfs = [], []
        for b in range(B):
            nf, _, nv = gt_moment_translate(fs[b], int(nfeats[b]), video[b:b + 1].double().numpy(), rng=rs)
            pv.append(torch.from_numpy(np.ascontiguousarray(nv[0])).float()); pfs.append([int(nf[0]), int(nf[1])])
        out["pseudo_video"] = torch.stack(pv)
        out["pseudo_gt"] = labels(pfs)
    if device != "cpu":
        for k, v in list(out.items()):
            if isinstance(v, torch.Tensor):
                out[k] = v.to(device)
        for gt in ("gt", "pseudo_gt"):
            if gt in out:
                for k, v in list(out[gt].items()):
                    if isinstance(v, torch.Tensor):
                        out[gt][k] = v.to(device)
                # the collate's list of [start, end] pairs becomes ONE resident index tensor: the losses gather with it
                # three times per step, and a list would be a blocking host-to-device copy each time
                out[gt]["framestps"] = torch.tensor(out[gt]["framestps"], dtype=torch.long).to(device)
    return out
