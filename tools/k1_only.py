#!/usr/bin/env python3
"""Launch K1 forward / backward N times at the north-star shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, N, d = 64, 128, 20, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev)
w = torch.randn(d, device=dev) / d ** 0.5; V = torch.randn(B, N, d, device=dev)
C = torch.empty(B, T, d, device=dev); P = torch.empty(B, T, N, device=dev); dC = torch.randn(B, T, d, device=dev)
da = torch.empty_like(A); ds = torch.empty_like(S); dw = torch.empty_like(w); dV = torch.empty_like(V); de = torch.empty(B, T, N, device=dev)
for _ in range(n):
    lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(C), ptr(P), B, T, N, d, d, TSG_F32, st)
    lib.tsg_scdm_attn_bwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dV), ptr(de), B, T, N, d, d, TSG_F32, st)
torch.cuda.synchronize()
print("done")
