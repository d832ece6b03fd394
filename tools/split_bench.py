#!/usr/bin/env python3
"""Bandwidth of the operand-split kernels (developer tool): plain vs transposing, at the LSTM weight-gradient shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as TF
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for R, C in [(16384, 4096), (16384, 1024), (16384, 512)]:
    x = torch.randn(R, C, device="cuda")
    buf = torch.empty(C, 3 * R, device="cuda", dtype=torch.bfloat16)
    nbytes = R * C * 10
    a = t(lambda: TF.split_bf16x3(x, 0, False))
    b = t(lambda: TF.split_bf16x3_t(x, 0, C, 0, False, buf))
    print(f"tile {os.environ.get('TSG_SPLIT_T_TILE', '0')} [{R}x{C}] plain {a:6.1f} us ({nbytes / a / 1e6:.2f} TB/s)   transposing {b:6.1f} us ({nbytes / b / 1e6:.2f} TB/s)")
