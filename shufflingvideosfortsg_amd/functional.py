"""torch.autograd.Function wrappers over the C ABI (one per fused kernel pair).

All inputs must be CUDA(HIP) fp32 tensors; they are made contiguous here; outputs and workspaces
are allocated from torch's caching allocator and handed to the library as raw pointers; the
kernels run on torch's current stream.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import TSG_F32, check, load, ptr, require_device, stream_of


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"fp32 tensor expected, got {t.dtype}")
    return t.contiguous()


class _ScdmAttn(torch.autograd.Function):
    """K1: (a=[B,T,H], s=[B,N,H], w=[H], sent=[B,N,Ds]) -> (C=[B,T,Ds], P=[B,T,N])."""

    @staticmethod
    def forward(ctx, a, s, w, sent):
        require_device(a, s, w, sent)
        a, s, w, sent = _f32c(a), _f32c(s), _f32c(w.reshape(-1)), _f32c(sent)
        B, T, H = a.shape
        _, N, Ds = sent.shape
        if s.shape != (B, N, H) or w.numel() != H:
            raise ValueError(f"scdm_attn: shape mismatch a{tuple(a.shape)} s{tuple(s.shape)} w{tuple(w.shape)} sent{tuple(sent.shape)}")
        C = torch.empty(B, T, Ds, device=a.device, dtype=torch.float32)
        P = torch.empty(B, T, N, device=a.device, dtype=torch.float32)
        check(load().tsg_scdm_attn_fwd(ptr(a), ptr(s), ptr(w), ptr(sent), ptr(C), ptr(P),
                                       B, T, N, H, Ds, TSG_F32, stream_of(a)), "tsg_scdm_attn_fwd")
        ctx.save_for_backward(a, s, w, sent, P)
        ctx.mark_non_differentiable(P)
        return C, P

    @staticmethod
    def backward(ctx, dC, _dP):
        a, s, w, sent, P = ctx.saved_tensors
        dC = _f32c(dC)
        B, T, H = a.shape
        _, N, Ds = sent.shape
        da = torch.empty_like(a); ds = torch.empty_like(s)
        dw = torch.empty_like(w); dsent = torch.empty_like(sent)
        de = torch.empty(B, T, N, device=a.device, dtype=torch.float32)
        check(load().tsg_scdm_attn_bwd(ptr(a), ptr(s), ptr(w), ptr(sent), ptr(P), ptr(dC), ptr(da), ptr(ds),
                                       ptr(dw), ptr(dsent), ptr(de), B, T, N, H, Ds, TSG_F32, stream_of(a)),
              "tsg_scdm_attn_bwd")
        return da, ds, dw, dsent


def scdm_attn(a, s, w, sent, return_p: bool = False):
    """Fused SCDM additive cross-attention on projected inputs (see include/tsg_hip.h, K1)."""
    C, P = _ScdmAttn.apply(a, s, w, sent)
    return (C, P) if return_p else C
