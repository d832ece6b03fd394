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
da = torch.empty_like(A); ds = torch.empty_like(S); dw = torch.empty_like(w); dV = torch.empty_like(V); VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev); r = torch.randn(B, T, d, device=dev)
out = torch.empty(B, T, d, device=dev); dVW = torch.empty_like(VW); dgb = torch.empty_like(gb); dr = torch.empty_like(r); nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 1)); ws = torch.empty(nb // 4 + 4, device=dev)
for _ in range(n):
    lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, TSG_F32, st)
    lib.tsg_scdm_gate_bwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
torch.cuda.synchronize()
for _ in range(n):
    lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(C), ptr(P), B, T, N, d, d, TSG_F32, st)
    lib.tsg_scdm_attn_bwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dV), ptr(ws), nb, B, T, N, d, d, TSG_F32, st)
torch.cuda.synchronize()
print("done")
