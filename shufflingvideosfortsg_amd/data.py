"""Input-side helpers of the grounding path: label masks, the shuffling augmentation and the
synthetic batches used by the benchmarks.

Two implementations of the same functions:
  * host / numpy, like the reference's dataset code (grounding/dataset/charades.py, data_augment.py): ``Sequence_mask``,
    ``gt_moment_translate``, ``synthetic_batch`` -- used for CPU tensors, by the CPU-side tests and to seed the benchmarks;
  * device (``*_device``, csrc/input_pipeline.hip through the C ABI): pair-mean pooling, the four masks and the shuffling
    augmentation for a whole batch on the GPU that consumes it, so that N ranks on one host do not queue behind numpy
    workers (the reference feeds ONE GPU from 8 DataLoader workers).  ``pair_batch_device`` assembles the GMD batch dict
    (train.py:50-91 keys) from resident features."""
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
    return out if device == "cpu" else to_device(out, device)


def to_device(batch, device):
    """Move a (possibly sharded) host batch dict to ``device``; the collate's lists of [start, end] pairs become ONE resident
    index tensor per label group: the losses gather with it three times per step, and a list would be a blocking host-to-device
    copy each time."""
    out = dict(batch)
    for k, v in list(out.items()):
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device)
    for gt in ("gt", "pseudo_gt"):
        if gt in out:
            out[gt] = dict(out[gt])
            for k, v in list(out[gt].items()):
                if isinstance(v, torch.Tensor):
                    out[gt][k] = v.to(device)
            fs = out[gt]["framestps"]
            out[gt]["framestps"] = (fs if isinstance(fs, torch.Tensor) else torch.tensor(fs, dtype=torch.long)).to(device)
    # The GMD step runs the original and the shuffled stream as ONE batch of 2B (model/SpanGroundMatchDisc.py): their tensors are
    # placed back to back in one device allocation, so the model's batch concatenation is a view (`adjacent_cat`), not a 64 MB copy
    # and four small ones per step.
    if "pseudo_video" in out and isinstance(out.get("video"), torch.Tensor):
        out["video"], out["pseudo_video"] = _adjacent(out["video"], out["pseudo_video"])
    if "gt" in out and "pseudo_gt" in out:
        for k in ("temporal_labels", "fore_masks", "back_masks"):
            a, b = out["gt"].get(k), out["pseudo_gt"].get(k)
            if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
                out["gt"][k], out["pseudo_gt"][k] = _adjacent(a, b)
    return out


def _adjacent(a, b):
    """Two equally shaped tensors as the two halves of one allocation (views)."""
    if a.shape != b.shape or a.dtype != b.dtype or a.device != b.device:
        return a, b
    both = torch.empty((2 * a.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    both[:a.shape[0]].copy_(a); both[a.shape[0]:].copy_(b)
    return both[:a.shape[0]], both[a.shape[0]:]


def adjacent_cat(a, b):
    """``torch.cat([a, b], 0)``; a VIEW when b starts where a ends in the same storage (see ``to_device``) and neither needs a gradient."""
    if (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
            and a.dim() >= 1 and a.is_contiguous() and b.is_contiguous() and not a.requires_grad and not b.requires_grad
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.storage_offset() + a.numel()
            and a.untyped_storage().nbytes() >= (b.storage_offset() + b.numel()) * a.element_size()):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], 0)


# ---------------------------------------------------------------------------------------------------------------------
# device-side pipeline (csrc/input_pipeline.hip); no CPU fallback: CPU tensors raise
# ---------------------------------------------------------------------------------------------------------------------

def _i32(t, device):
    return torch.as_tensor(t, device=device).to(torch.int32).contiguous()


def pool_clips_device(raw, offsets, T, timestamps=None):
    """Charades pair-mean pooling + zero pad of a batch (generate_video_fts_data, charades.py:177-196): ``raw`` [sum n_b, D]
    fp32 device tensor (the batch's clip features back to back), ``offsets`` [B+1] -> (video [B,T,D], nfeats [B] int32,
    framestps [B,2] int32 or None when no ``timestamps`` [B,2] (seconds) are given)."""
    from . import _lib
    from .functional import _call
    _lib.require_device(raw)
    raw = raw.contiguous()
    off = torch.as_tensor(offsets, device=raw.device).to(torch.int64).contiguous()
    B, D = off.numel() - 1, raw.shape[1]
    out = torch.empty(B, T, D, device=raw.device, dtype=torch.float32)
    nf = torch.empty(B, device=raw.device, dtype=torch.int32)
    ts = fs = None
    if timestamps is not None:
        ts = torch.as_tensor(timestamps, device=raw.device).to(torch.float64).contiguous()
        fs = torch.empty(B, 2, device=raw.device, dtype=torch.int32)
    _call("tsg_pool_clips", raw, raw.data_ptr(), off.data_ptr(), ts.data_ptr() if ts is not None else None, out.data_ptr(),
          nf.data_ptr(), fs.data_ptr() if fs is not None else None, B, T, D, _lib.TSG_F32)
    return out, nf, fs


def sequence_masks_device(nfeats, spans, T):
    """-> dict(video_mask, temporal_labels, fore_masks, back_masks), each int32 [B,T] (Sequence_mask as combined at
    charades.py:167-170) from device tensors nfeats [B] and spans [B,2]."""
    from . import _lib
    from .functional import _call
    _lib.require_device(nfeats, spans)
    nf, sp = _i32(nfeats, nfeats.device), _i32(spans, nfeats.device)
    B = nf.numel()
    vm, tl, fm, bm = (torch.empty(B, T, device=nf.device, dtype=torch.int32) for _ in range(4))
    _call("tsg_sequence_masks", nf, nf.data_ptr(), sp.data_ptr(), vm.data_ptr(), tl.data_ptr(), fm.data_ptr(), bm.data_ptr(), B, T)
    return {"video_mask": vm, "temporal_labels": tl, "fore_masks": fm, "back_masks": bm}


def gt_moment_translate_device(video, spans, nfeats, cropin_start=None, seed=0):
    """The shuffling augmentation for a whole batch on the device (gt_moment_translate, data_augment.py:135-156):
    video [B,T,D] fp32, spans [B,2], nfeats [B] -> (pseudo video [B,T,D], new spans [B,2] int32).  ``cropin_start`` [B]:
    insert positions (the reference draws them with random.randint(0, nfeats-len)); None: drawn on the device from a
    counter-based hash of (seed, sample index)."""
    from . import _lib
    from .functional import _call
    _lib.require_device(video)
    video = video.contiguous()
    B, T, D = video.shape
    sp, nf = _i32(spans, video.device), _i32(nfeats, video.device)
    ci = _i32(cropin_start, video.device) if cropin_start is not None else None
    out = torch.empty_like(video)
    new = torch.empty(B, 2, device=video.device, dtype=torch.int32)
    _call("tsg_moment_translate", video, video.data_ptr(), sp.data_ptr(), nf.data_ptr(), ci.data_ptr() if ci is not None else None,
          int(seed) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), new.data_ptr(), B, T, D, _lib.TSG_F32)
    return out, new


def cropin_positions(seed, spans, nfeats):
    """Host mirror of the device's insert-position draw (splitmix64 of (seed, b), multiply-high onto [0, nfeats-len]) --
    what ``gt_moment_translate_device(..., cropin_start=None, seed=seed)`` uses; for tests and for reproducing a batch."""
    M = (1 << 64) - 1

    def mix(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    out = []
    for b, ((s, e), nf) in enumerate(zip(np.asarray(spans).tolist(), np.asarray(nfeats).tolist())):
        n = e - s + 1
        if n <= 1 or n >= nf:
            out.append(s)
            continue
        h = mix((int(seed) & M) ^ ((0xD1B54A32D192ED03 * (b + 1)) & M))
        out.append(((h >> 32) * (nf - n + 1)) >> 32)
    return out


def pair_batch_device(video, nfeats, spans, query, seed=0, cropin_start=None, timestps=None):
    """The GMD training batch (keys of train.py:50-91 / charades_pair_aug.py:96-107) assembled on the device from resident
    tensors: video [B,T,D], nfeats [B], spans [B,2] (frame stamps), query [B,N,300].  Masks and the shuffled pseudo video
    come from the device kernels; nothing is copied to or from the host."""
    B, T, _ = video.shape
    sp, nf = _i32(spans, video.device), _i32(nfeats, video.device)
    pv, psp = gt_moment_translate_device(video, sp, nf, cropin_start, seed)
    m, pm = sequence_masks_device(nf, sp, T), sequence_masks_device(nf, psp, T)
    out = {"video": video, "query": query, "nfeats": nf, "video_mask": m.pop("video_mask"),
           "query_mask": torch.ones(query.shape[0], query.shape[1], dtype=torch.int32, device=video.device),
           "pseudo_video": pv, "gt": m, "pseudo_gt": pm}
    pm.pop("video_mask")                      # gt_translate keeps nfeats: the pseudo video shares the mask (engine.gmd_step)
    out["gt"]["framestps"], out["pseudo_gt"]["framestps"] = sp.long(), psp.long()
    out["gt"]["timestps"] = timestps if timestps is not None else sp.float()
    return out
