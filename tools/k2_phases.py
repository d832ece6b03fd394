#!/usr/bin/env python3
"""Developer tool: s_memtime phase sums of the split-precision dK/dV kernel (needs -DTSG_K2_TIMING: TSG_VARIANT_SRC=mha tools/build_variant.sh k2time -DTSG_K2_TIMING)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import TSG_F32, TSG_F32S, ptr
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
B, T, Tk, d, h = 64, 128, 128, 1024, 8
Q = torch.randn(B, T, d, device="cuda"); K = torch.randn(B, Tk, d, device="cuda"); V = torch.randn(B, Tk, d, device="cuda")
O = torch.empty(B, T, d, device="cuda"); lse = torch.empty(B, h, T, device="cuda"); g = torch.randn(B, T, d, device="cuda")
dQ, dK, dV, dlt = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(lse)
sc = math.sqrt(d)
lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32, st)
for _ in range(3):
    lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32S, st)
torch.cuda.synchronize()
v = dK.view(-1)[:32].cpu().view(4, 8)
print("cycles (whole kernel, 4 tiles): prologue | request | S,dP | softmax | dV,dK | stage | barrier | epilogue")
for w in range(4):
    print(f" wave {w}: " + " ".join(f"{float(x):8.0f}" for x in v[w]) + f" | sum {float(v[w].sum()):8.0f}")
