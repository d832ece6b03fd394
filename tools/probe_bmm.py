#!/usr/bin/env python3
"""Weight-gradient GEMM as one merged GEMM (with the two cross-direction blocks) vs a 2-batch bmm without them (probe)."""
import torch
TB3, h4, I, h = 3 * 16384, 2048, 1024, 512
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
A = (torch.randn(2 * h4, TB3, device="cuda") * 0.1).bfloat16()
Bm = (torch.randn(I + 2 * h, TB3, device="cuda") * 0.1).bfloat16()
Bb = (torch.randn(2, I + h, TB3, device="cuda") * 0.1).bfloat16()
print(f"merged [4096 x {I + 2 * h}]: {t(lambda: torch.mm(A, Bm.t(), out_dtype=torch.float32)):.1f} us")
try:
    print(f"bmm 2 x [2048 x {I + h}]: {t(lambda: torch.bmm(A.view(2, h4, TB3), Bb.transpose(1, 2), out_dtype=torch.float32)):.1f} us")
except Exception as e:
    print("bmm out_dtype not available:", str(e)[:200])
    print(f"bmm bf16 out: {t(lambda: torch.bmm(A.view(2, h4, TB3), Bb.transpose(1, 2))):.1f} us")
