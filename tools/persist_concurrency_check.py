#!/usr/bin/env python3
"""The persistent LSTM kernels must survive another stream's kernels holding CUs when they start (as an RCCL bucket
all-reduce would during backward): same results, no tripped wait, bounded delay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import functional as TF
T, B, I, h = 128, 128, 1024, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, I, generator=g).cuda().requires_grad_(True)
W_ih = (torch.randn(8 * h, I, generator=g) / I ** 0.5).cuda().requires_grad_(True)
bias = (torch.randn(8 * h, generator=g) * 0.1).cuda().requires_grad_(True)
W_hh = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda().requires_grad_(True)
go = torch.randn(T, B, 2 * h, generator=g).cuda()
side = torch.cuda.Stream(); big = torch.randn(256, 1024, 1024, device="cuda")
def run(disturb):
    for t in (x, W_ih, bias, W_hh): t.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if disturb:
        with torch.cuda.stream(side):
            for _ in range(6): big.sin_()
    out, _ = TF.bilstm_layer(x, W_ih, bias, W_hh)
    if disturb:
        with torch.cuda.stream(side):
            for _ in range(6): big.cos_()
    out.backward(go)
    torch.cuda.synchronize()
    return out.detach().clone(), [t.grad.clone() for t in (x, W_ih, bias, W_hh)], (time.perf_counter() - t0) * 1e3
run(False)
o0, g0, t0 = run(False)
o1, g1, t1 = run(True)
print(f"undisturbed {t0:.2f} ms, with a busy side stream {t1:.2f} ms")
print("outputs equal:", torch.equal(o0, o1), " grads max diff:", max((a - b).abs().max().item() for a, b in zip(g0, g1)), " finite:", all(torch.isfinite(t).all().item() for t in g1))
