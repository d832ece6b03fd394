#!/usr/bin/env python3
"""Does replaying the 128 backward step launches from a HIP graph shrink the launch gaps?  (developer probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, h = 128, 128, 512
lib = _lib.load(); dev = "cuda"
Gx = torch.randn(T, B, 2, 4 * h, device=dev) * 0.5; W = torch.randn(2, 4 * h, h, device=dev) / h ** 0.5
out = torch.empty(T, B, 2 * h, device=dev); R = torch.empty(T, 2, B, h, 4, device=dev); Cs = torch.empty(T, 2, B, h, device=dev)
dOut = torch.randn(T, B, 2 * h, device=dev); WT = W.transpose(1, 2).contiguous()
dG = torch.empty(T, B, 2, 4 * h, device=dev); dC = torch.empty(2, B, h, device=dev)
lib.tsg_lstm_fwd(ptr(Gx), ptr(W), ptr(out), ptr(R), ptr(Cs), None, B, T, h, TSG_F32, torch.cuda.current_stream().cuda_stream)
def bwd(): lib.tsg_lstm_bwd(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), B, T, h, TSG_F32, torch.cuda.current_stream().cuda_stream)
def wall(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(f"eager: {wall(bwd):.3f} ms per layer ({wall(bwd) * 1e3 / T:.2f} us/step)")
ref = dG.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    bwd(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        bwd()
torch.cuda.synchronize()
dG.zero_(); g.replay(); torch.cuda.synchronize()
print("graph replay matches eager:", torch.equal(dG, ref))
print(f"graph: {wall(g.replay):.3f} ms per layer ({wall(g.replay) * 1e3 / T:.2f} us/step)")
