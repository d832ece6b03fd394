#!/usr/bin/env python3
"""K5 (matching head) forward / backward time at the bench shape (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as TF
B, T, H = 128, 128, 1024
y = torch.randn(B, T, H, device="cuda", requires_grad=True); cs = torch.randn(B, H, device="cuda", requires_grad=True)
w2 = (torch.randn(H, device="cuda") / 32).requires_grad_(True); b2 = torch.zeros(1, device="cuda", requires_grad=True); g = torch.randn(B, T, device="cuda")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = t(lambda: TF.match_head(y, cs, w2, b2, "relu"))
def fb():
    out = TF.match_head(y, cs, w2, b2, "relu"); torch.autograd.grad(out, (y, cs, w2, b2), g)
fbt = t(fb)
print(f"K5 [{B},{T},{H}]: forward {f:.1f} us ({B*T*H*4/f/1e6:.2f} TB/s), forward+backward {fbt:.1f} us (backward ~{fbt-f:.1f} us, {2*B*T*H*4/(fbt-f)/1e6:.2f} TB/s)")
