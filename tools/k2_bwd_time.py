#!/usr/bin/env python3
"""Developer tool: tsg_mha_bwd exact-fp32 vs split-precision (TSG_F32S) kernels, back-to-back launches timed with events."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import TSG_F32, TSG_F32S, ptr
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (B, T, Tk, d, h) in [(64, 128, 128, 1024, 8), (64, 128, 20, 1024, 8), (64, 128, 128, 512, 8), (16, 512, 512, 1024, 8), (64, 128, 128, 2048, 8)]:
    Q = torch.randn(B, T, d, device="cuda"); K = torch.randn(B, Tk, d, device="cuda"); V = torch.randn(B, Tk, d, device="cuda")
    O = torch.empty(B, T, d, device="cuda"); lse = torch.empty(B, h, T, device="cuda"); g = torch.randn(B, T, d, device="cuda")
    dQ, dK, dV, dlt = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(lse)
    sc = math.sqrt(d)
    lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, TSG_F32, st)
    fres = []
    for dt in (TSG_F32, TSG_F32S):
        fn = lambda: lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, dt, st)
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); e1.synchronize()
        fres.append(e0.elapsed_time(e1) * 1e3 / 30)
    fby = B * (2 * T + 2 * Tk) * d * 4
    print(f"[{B},{T},{Tk},{d},h{h}] forward : fp32 kernels {fres[0]:7.1f} us, split-precision {fres[1]:7.1f} us = {fby / fres[1] / 1e3 / 8000:.3f} of the HBM roofline")
    res = []
    for dt in (TSG_F32, TSG_F32S):
        fn = lambda: lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d, h, sc, 0, 0.0, 0, 0, dt, st)
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 30)
    by = B * (4 * T + 4 * Tk) * d * 4
    print(f"[{B},{T},{Tk},{d},h{h}] backward: fp32 kernels {res[0]:7.1f} us, split-precision {res[1]:7.1f} us = {by / res[1] / 1e3 / 8000:.3f} of the HBM roofline")
