#!/usr/bin/env python3
"""tsg_wgrad_bf16 vs the library's bf16 GEMM (fp32 output) at the weight-gradient shapes of the bf16 storage step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F


def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((16384, 1024, 1024), (16384, 512, 1024), (2560, 1024, 1024), (16384, 1024, 2048)):
    A = torch.randn(M, N, device="cuda").bfloat16(); B = torch.randn(M, K, device="cuda").bfloat16()
    t1 = timeit(lambda: F.wgrad_bf16(A, B)); t2 = timeit(lambda: torch.mm(A.t(), B, out_dtype=torch.float32))
    fl = 2.0 * M * N * K
    print(f"dW [{N}x{M}].[{M}x{K}]: tsg_wgrad_bf16 {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TFLOP/s)   torch.mm {t2:7.1f} us ({fl / t2 / 1e6:6.0f} TFLOP/s)")
# the LSTM layer's weight gradients, both directions, one launch vs three GEMMs
TB, h, I, T = 16384, 512, 1024, 128
dG = torch.randn(TB, 8 * h, device="cuda").bfloat16(); x = torch.randn(TB, I, device="cuda").bfloat16(); out = torch.randn(TB, 2 * h, device="cuda").bfloat16()
t1 = timeit(lambda: F.wgrad_bf16(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=1, period=T))
g2 = dG.view(TB, 2, 4 * h)
def lib():
    torch.mm(dG.t(), x, out_dtype=torch.float32)
    torch.mm(g2[:, 0].t(), out[:, :h], out_dtype=torch.float32); torch.mm(g2[:, 1].t(), out[:, h:], out_dtype=torch.float32)
t2 = timeit(lib)
print(f"LSTM layer dW_ih + dW_hh (h=512, I=1024, T*B=16384): one tsg_wgrad_bf16 launch {t1:7.1f} us   three library GEMMs {t2:7.1f} us")
