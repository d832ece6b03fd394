#!/usr/bin/env python3
"""Does the library's bf16 GEMM run the weight-gradient shape faster with K-contiguous operands? (developer probe)"""
import torch
import sys
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 2048, 3 * 16384)
a_km = torch.randn(K, M, device="cuda").bfloat16()      # [K, M]: K is the slow dimension (what split_bf16x3(k_dim=0) writes)
b_kn = torch.randn(K, N, device="cuda").bfloat16()
a_mk = a_km.t().contiguous()                             # [M, K]: K contiguous
b_nk = b_kn.t().contiguous()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * M * N * K
for name, fn in [("TN  a[K,M]^T b[K,N]  ", lambda: torch.mm(a_km.t(), b_kn, out_dtype=torch.float32)),
                 ("NT  a[M,K]   b[N,K]^T", lambda: torch.mm(a_mk, b_nk.t(), out_dtype=torch.float32)),
                 ("NN  a[M,K]   b[K,N]  ", lambda: torch.mm(a_mk, b_kn, out_dtype=torch.float32)),
                 ("TT  a[K,M]^T b[N,K]^T", lambda: torch.mm(a_km.t(), b_nk.t(), out_dtype=torch.float32))]:
    us = t(fn)
    print(f"{name}: {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s")
# per-direction variant without the wasted cross blocks: 2 x [2048 x 1536]
for name, fn in [] if M != 4096 else [("NT per direction x2  ", lambda: (torch.mm(a_mk[:2048], b_nk[:1536].t(), out_dtype=torch.float32), torch.mm(a_mk[2048:], b_nk[512:].t(), out_dtype=torch.float32)))]:
    us = t(fn)
    print(f"{name}: {us:7.1f} us")
