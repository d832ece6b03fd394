#!/usr/bin/env python3
"""Event-timed K2 (multi-head attention) forward / backward at the north-star shape: cross (Tk=20) and self (Tk=128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, heads = 64, 128, 8
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tag, Tk in (("cross", 20), ("self", 128)):
    Q = torch.randn(B, T, d, device=dev); K = torch.randn(B, Tk, d, device=dev); V = torch.randn(B, Tk, d, device=dev)
    O = torch.empty(B, T, d, device=dev); lse = torch.empty(B, heads, T, device=dev); g = torch.randn(B, T, d, device=dev)
    dQ = torch.empty_like(Q); dK = torch.empty_like(K); dV = torch.empty_like(V); dlt = torch.empty(B, heads, T, device=dev)
    sc = d ** 0.5
    f = timeit(lambda: lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, heads, sc, 0, 0.0, 0, 0, TSG_F32, st))
    b = timeit(lambda: lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, heads, sc, 0, 0.0, 0, 0, TSG_F32, st))
    fb = B * (2 * T + 2 * Tk) * d * 4; bb = B * (4 * T + 4 * Tk) * d * 4
    print(f"d={d} h{heads} {tag}: fwd {f:.1f} us ({fb/f/1e3/8000*100:.1f}% of 8 TB/s)   bwd {b:.1f} us ({bb/b/1e3/8000*100:.1f}%)")
