#!/usr/bin/env python3
"""Do library GEMMs on a side stream overlap with the latency-bound LSTM step kernels?  (developer probe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, h = 128, 128, 512
lib = _lib.load(); dev = "cuda"
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
Gx = torch.randn(T, B, 2, 4 * h, device=dev) * 0.5; W = torch.randn(2, 4 * h, h, device=dev) / h ** 0.5
out = torch.empty(T, B, 2 * h, device=dev); R = torch.empty(T, 2, B, h, 4, device=dev); Cs = torch.empty(T, 2, B, h, device=dev)
dOut = torch.randn(T, B, 2 * h, device=dev); WT = W.transpose(1, 2).contiguous()
dG = torch.empty(T, B, 2, 4 * h, device=dev); dC = torch.empty(2, B, h, device=dev)
sync = torch.zeros(512, device=dev, dtype=torch.int32)
lib.tsg_lstm_fwd(ptr(Gx), ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, TSG_F32, main.cuda_stream)
a = torch.randn(4096, 16384, device=dev); x = torch.randn(16384, 1024, device=dev); c = torch.empty(4096, 1024, device=dev)
def lstm_b(): lib.tsg_lstm_bwd(ptr(WT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), B, T, h, TSG_F32, main.cuda_stream)
def lstm_f(): lib.tsg_lstm_fwd(ptr(Gx), ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, TSG_F32, main.cuda_stream)
def gemms(n=2):
    for _ in range(n): torch.mm(a, x, out=c)
def wall(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for name, lstm in (("bwd step kernels", lstm_b), ("fwd persistent", lstm_f)):
    t_l = wall(lstm); t_g = wall(gemms)
    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side): gemms()
        lstm()
        main.wait_stream(side)
    t_b = wall(both)
    print(f"{name}: lstm {t_l:.2f} ms, 2 GEMMs {t_g:.2f} ms, serial sum {t_l + t_g:.2f} ms, two streams {t_b:.2f} ms")
