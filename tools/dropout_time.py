#!/usr/bin/env python3
"""Developer tool: tsg_dropout vs torch's native_dropout, forward + backward, at the inter-layer shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from shufflingvideosfortsg_amd import functional as F

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for dt in (torch.float32, torch.bfloat16):
    for shape in ((128, 128, 1024), (64, 20, 1024)):
        x = torch.randn(*shape, device="cuda").to(dt).requires_grad_(True)
        g = torch.randn(*shape, device="cuda").to(dt)
        def own():
            x.grad = None; F.dropout(x, 0.5).backward(g)
        def ref():
            x.grad = None; TF.dropout(x, 0.5, True).backward(g)
        def own_f():
            with torch.no_grad(): F._Dropout.apply(x, 0.5, 1, 2, None)
        def ref_f():
            with torch.no_grad(): TF.dropout(x, 0.5, True)
        print(f"{dt} {shape}: fwd+bwd own {t(own):.1f} us, torch {t(ref):.1f} us; fwd only own {t(own_f):.1f} us, torch {t(ref_f):.1f} us")
