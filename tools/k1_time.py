#!/usr/bin/env python3
"""Event-timed K1 gate-fused / plain forward at B pairs per launch: python tools/k1_time.py [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T, N, d = 128, int(sys.argv[3]) if len(sys.argv) > 3 else 20, 1024
lib = _lib.load(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev)
w = torch.randn(d, device=dev) / d ** 0.5; V = torch.randn(B, N, d, device=dev)
C = torch.empty(B, T, d, device=dev); P = torch.empty(B, T, N, device=dev)
VW = torch.randn(B, N, d, device=dev); gb = torch.randn(d, device=dev); r = torch.randn(B, T, d, device=dev)
out = torch.empty(B, T, d, device=dev)
def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = timeit(lambda: lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, TSG_F32, st))
p = timeit(lambda: lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(C), ptr(P), B, T, N, d, d, TSG_F32, st))
bg = ((3 * T + 2 * N) * d * 4 + T * N * 4) * B; bp = ((2 * T + 2 * N) * d * 4 + T * N * 4) * B
print(f"B={B} TT={os.environ.get('TSG_K1_TT','auto')}: gate fwd {g:.1f} us ({bg/g/1e3/8000*100:.1f}% of 8 TB/s)   plain fwd {p:.1f} us ({bp/p/1e3/8000*100:.1f}%)")
