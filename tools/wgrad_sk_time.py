#!/usr/bin/env python3
"""Developer tool: the weight-gradient kernels at the step's shapes (tsg_wgrad_f32s_out2 / tsg_wgrad_bf16_out2 for an LSTM layer, tsg_wgrad_f32s for
the projections), by events over 20 launches.  TSG_WGRAD_SK=0 in the environment = the row-range scheme + reduce kernel (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as F
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
Bn, T, h = 128, 128, 512
for I in (1024,):
    TB = Bn * T
    dG = torch.randn(TB, 8 * h, generator=g).cuda(); x = torch.randn(TB, I, generator=g).cuda(); out = torch.randn(TB, 2 * h, generator=g).cuda()
    t = timeit(lambda: F.wgrad_f32s_out2(dG, x, out, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=1, period=T))
    fl = 2.0 * 2 * 4 * h * (I + h) * TB * 3
    print(f"f32s LSTM layer dW [2][{4*h} x {TB}] x [{TB} x {I + h}]: {t:7.1f} us  {fl / t / 1e9:.3f} PFLOP/s of bf16 MFMA work")
    dGb, xb, ob = dG.bfloat16(), x.bfloat16(), out.bfloat16()
    t = timeit(lambda: F.wgrad_bf16_out2(dGb, xb, ob, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=1, period=T))
    print(f"bf16 LSTM layer dW: {t:7.1f} us  {fl / 3 / t / 1e9:.3f} PFLOP/s")
for (M, N, K) in ((16384, 1024, 1024), (16384, 1024, 2048), (16384, 512, 2048), (1280, 2048, 1536)):
    A = torch.randn(M, N, generator=g).cuda(); B = torch.randn(M, K, generator=g).cuda()
    if not F.wgrad_f32s_ok(M, N, K, 0):
        print(f"[{N} x {M}] x [{M} x {K}]: not taken by the kernel"); continue
    t = timeit(lambda: F.wgrad_f32s(A, B))
    print(f"f32s [{N} x {M}] x [{M} x {K}]: {t:7.1f} us  {2.0 * M * N * K * 3 / t / 1e9:.3f} PFLOP/s")
