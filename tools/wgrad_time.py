#!/usr/bin/env python3
"""Developer tool: tsg_wgrad_f32s vs the paths it replaces, by HIP events.   python tools/wgrad_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (M, N, K) in [(16384, 1024, 1024), (2560, 1024, 1024), (8192, 512, 1024), (16384, 2048, 2048), (8192, 2048, 2048), (16384, 256, 1024)]:
    A, B = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
    t_new = timeit(lambda: F.wgrad_f32s(A, B))
    t_f32 = timeit(lambda: A.t() @ B)

    def lib():
        At = torch.empty(N, 3 * M, device="cuda", dtype=torch.bfloat16); Bt = torch.empty(K, 3 * M, device="cuda", dtype=torch.bfloat16)
        F.split_bf16x3_t(A, 0, N, 0, False, At); F.split_bf16x3_t(B, 0, K, 0, True, Bt)
        return torch.mm(At, Bt.t(), out_dtype=torch.float32)
    t_lib = timeit(lib)
    fl = 3 * 2.0 * M * N * K
    print(f"[{N}x{M}]x[{M}x{K}]: wgrad_f32s {t_new:7.1f} us ({fl / t_new / 1e9:6.1f} PFLOP/s x1e-3 of bf16 work)  fp32 GEMM {t_f32:7.1f} us  split_t+bf16 GEMM {t_lib:7.1f} us")

# LSTM layer shape: dG [16384, 4096], x [16384, 1024], out [16384, 1024]
TB, I, h, T = 16384, 1024, 512, 128
dG, x, out = torch.randn(TB, 8 * h, device="cuda"), torch.randn(TB, I, device="cuda"), torch.randn(TB, 2 * h, device="cuda")
t_new = timeit(lambda: F.wgrad_f32s(dG, x, N=4 * h, groups=2, a_group_stride=4 * h, B1=out, K1=h, b1_group_stride=h, shift=1, period=T))


def lib_lstm():
    At = torch.empty(8 * h, 3 * TB, device="cuda", dtype=torch.bfloat16)
    Bt = torch.empty(2 * (I + h), 3 * TB, device="cuda", dtype=torch.bfloat16)
    F.split_bf16x3_t(dG, 0, 8 * h, 0, False, At)
    F.split_bf16x3_t(x, 0, I, 0, True, Bt, 0, dup_row0=I + h)
    F.split_bf16x3_t(out, 0, h, 1, True, Bt, I, T)
    F.split_bf16x3_t(out, h, h, -1, True, Bt, 2 * I + h, T)
    return torch.bmm(At.view(2, 4 * h, 3 * TB), Bt.view(2, I + h, 3 * TB).transpose(1, 2), out_dtype=torch.float32)
t_lib = timeit(lib_lstm)
fl = 3 * 2.0 * TB * 8 * h * (I + h)
print(f"LSTM dW [2][{4*h}x{TB}]x[{TB}x{I+h}]: wgrad_f32s {t_new:7.1f} us ({fl / t_new / 1e9:6.1f})  split_t x4 + bmm {t_lib:7.1f} us")
