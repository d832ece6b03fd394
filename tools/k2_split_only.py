#!/usr/bin/env python3
"""Profiling target: n forward + backward launches of K2 in split precision (TSG_F32S) at one shape.  python tools/k2_split_only.py n B Tq Tk d heads"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import TSG_F32S, ptr
n, B, T, Tk, d, h = (int(x) for x in sys.argv[1:7])
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
Q = torch.randn(B, T, d, device="cuda"); K = torch.randn(B, Tk, d, device="cuda"); V = torch.randn(B, Tk, d, device="cuda")
O = torch.empty(B, T, d, device="cuda"); lse = torch.empty(B, h, T, device="cuda"); g = torch.randn(B, T, d, device="cuda")
dQ, dK, dV, dlt = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(lse)
sc = math.sqrt(d)
for _ in range(n):
    assert lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32S, st) == 0
    assert lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32S, st) == 0
torch.cuda.synchronize()
