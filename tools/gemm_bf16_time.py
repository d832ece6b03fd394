#!/usr/bin/env python3
"""Developer tool: tsg_gemm_bf16 (csrc/gemm_bf16.hip) vs torch.mm (hipBLASLt) at the bf16 step's shapes, by events over 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F
BF = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((16384, 4096, 1024), (16384, 1024, 4096), (16384, 1024, 1024), (16384, 1024, 2048), (16384, 512, 2048), (2560, 1024, 1024), (1280, 4096, 1024), (8192, 1024, 1024)):
    X = torch.randn(M, K, generator=g).to(BF).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).cuda()
    to = timeit(lambda: F.gemm_bf16(X, W))
    tl = timeit(lambda: torch.mm(X, W.t()))
    Wt = W.t().contiguous()
    tn = timeit(lambda: torch.mm(X, Wt))
    fl = 2.0 * M * N * K
    print(f"[{M} x {K}] . [{N} x {K}]^T: own {to:7.1f} us {fl / to / 1e9:.3f} PFLOP/s | hipBLASLt NT {tl:7.1f} us {fl / tl / 1e9:.3f} | NN {tn:7.1f} us")
