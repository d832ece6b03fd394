#!/usr/bin/env python3
"""tsg_linear_fwd (hand-written fp32 MFMA GEMM) vs rocBLAS F.linear: correctness (vs float64) and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for (M, N, K) in [(8192, 1024, 1024), (16384, 1024, 1024), (16384, 4096, 1024), (16384, 512, 1024), (1280, 1024, 1024), (100, 72, 36)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    rc = lib.tsg_linear_fwd(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, TSG_F32, st); torch.cuda.synchronize()
    assert rc == 0, lib.tsg_last_error()
    ref = (x.double() @ w.double().t() + b.double())
    e_h = (y.double() - ref).abs().max().item(); e_r = (torch.nn.functional.linear(x, w, b).double() - ref).abs().max().item()
    th = timeit(lambda: lib.tsg_linear_fwd(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, TSG_F32, st))
    tr = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"[{M}x{K}]x[{N}x{K}]^T  hip {th:7.1f} us {fl/th/1e6:6.1f} TF/s  rocBLAS {tr:7.1f} us {fl/tr/1e6:6.1f} TF/s   max err vs f64: hip {e_h:.2e} rocBLAS {e_r:.2e}")
