#!/usr/bin/env python3
"""Can the four GMD losses (forward + backward, ~100 tiny launches) run as two HIP-graph replays? (developer probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import loss as L
from shufflingvideosfortsg_amd.model.networks.attention import masked_softmax
dev = "cuda"; B, T = 64, 128
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
start = torch.softmax(mk(B, T), 1).requires_grad_(True); end = torch.softmax(mk(B, T), 1).requires_grad_(True)
om = mk(B, T).requires_grad_(True); pm = mk(B, T).requires_grad_(True); od = mk(B, 2).requires_grad_(True); pd = mk(B, 2).requires_grad_(True)
fs = torch.stack([torch.randint(0, 40, (B,), generator=g), torch.randint(50, 100, (B,), generator=g)], 1).to(dev)
pfs = fs + 5
tl = (torch.rand(B, T, generator=g) > 0.5).float().to(dev); ptl = (torch.rand(B, T, generator=g) > 0.5).float().to(dev)
vm = torch.ones(B, T, device=dev)
def loss_fn(start, end, om, pm, od, pd, fs, pfs, tl, ptl, vm):
    lg = L.span_ground_loss(start, end, fs)
    l1 = L.BCE_loss(om, tl, vm) + L.BCE_loss(pm, ptl, vm)
    l2 = L.matching_KL_divergence(masked_softmax(om, tl), masked_softmax(pm, ptl), fs, pfs)
    ld = L.temporal_order_discrimination_loss(od, pd)
    return lg + l1 + l2 + ld
args = (start, end, om, pm, od, pd, fs, pfs, tl, ptl, vm)
def run(fn, n=50):
    for _ in range(5):
        l = fn(*args); grads = torch.autograd.grad(l, args[:6])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        l = fn(*args); grads = torch.autograd.grad(l, args[:6])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, l.detach(), [x.detach() for x in grads]
t_e, l_e, g_e = run(loss_fn)
graphed = torch.cuda.make_graphed_callables(loss_fn, args)
t_g, l_g, g_g = run(graphed)
print(f"eager {t_e:.0f} us per loss fwd+bwd, graphed {t_g:.0f} us; loss {float(l_e):.6f} vs {float(l_g):.6f}; max grad diff {max(float((a - b).abs().max()) for a, b in zip(g_e, g_g)):.2e}")
