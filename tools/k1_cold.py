#!/usr/bin/env python3
"""K1g forward with COLD caches: a 1 GiB fill between launches evicts L2 and the 256 MiB Infinity Cache, so that a / r / s / VW come
from HBM as in the train step (stand-alone back-to-back launches keep the 201 MB working set cache-resident).  Events bracket the K1
launch only.   python tools/k1_cold.py [B] [dtype 0|1|2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, N, d = 128, 20, 1024
lib = _lib.load(); dev = "cuda"; stream = torch.cuda.current_stream(); st = stream.cuda_stream
big = torch.empty(1 << 28, device=dev)            # 1 GiB
for dt in ([int(sys.argv[2])] if len(sys.argv) > 2 else [0, 2, 1]):
    cast = torch.bfloat16 if dt == 1 else torch.float32
    A = torch.randn(B, T, d, device=dev).to(cast); S = torch.randn(B, N, d, device=dev).to(cast); w = torch.randn(d, device=dev) / 32
    VW = torch.randn(B, N, d, device=dev).to(cast); gb = torch.randn(d, device=dev); r = torch.randn(B, T, d, device=dev).to(cast)
    out = torch.empty(B, T, d, device=dev, dtype=cast); P = torch.empty(B, T, N, device=dev)
    fn = lambda: lib.tsg_scdm_gate_fwd(ptr(A), ptr(S), ptr(w), ptr(VW), ptr(gb), ptr(r), ptr(out), ptr(P), B, T, N, d, d, dt, st)
    for mode in ("warm", "cold"):
        ts = []
        for i in range(40):
            if mode == "cold":
                big.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fn(); e1.record(stream)
            ts.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[10:])
        print(f"dtype {dt} {mode}: median {v[len(v) // 2]:.1f} us  min {v[0]:.1f}  max {v[-1]:.1f}", flush=True)
