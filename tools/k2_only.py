#!/usr/bin/env python3
"""Launch K2 forward+backward a few times (for rocprofv3): python3 tools/k2_only.py [Tk] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
Tk = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, T, d, heads = 64, 128, 1024, 8
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
Q = torch.randn(B, T, d, device=dev); K = torch.randn(B, Tk, d, device=dev); V = torch.randn(B, Tk, d, device=dev)
O = torch.empty(B, T, d, device=dev); lse = torch.empty(B, heads, T, device=dev); g = torch.randn(B, T, d, device=dev)
dQ = torch.empty_like(Q); dK = torch.empty_like(K); dV = torch.empty_like(V); dlt = torch.empty(B, heads, T, device=dev)
sc = d ** 0.5
for _ in range(n):
    lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, heads, sc, 0, 0.0, 0, 0, TSG_F32, st)
    lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, heads, sc, 0, 0.0, 0, 0, TSG_F32, st)
torch.cuda.synchronize()
