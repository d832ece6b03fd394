#!/usr/bin/env python3
"""Compare the persistent LSTM forward (TSG_LSTM_PERSIST from the env) with the step kernels: B T h [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shufflingvideosfortsg_amd import _lib
from shufflingvideosfortsg_amd._lib import ptr, TSG_F32
B, T, h = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
Gx = (torch.randn(T, B, 2, 4 * h, generator=g) * 0.5).cuda(); W = (torch.randn(2, 4 * h, h, generator=g) / h ** 0.5).cuda()
def run(ws):
    out = torch.full((T, B, 2 * h), 7.0, device="cuda"); R = torch.empty(T, 2, B, h, 4, device="cuda"); Cs = torch.empty(T, 2, B, h, device="cuda")
    rc = lib.tsg_lstm_fwd(ptr(Gx), ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(ws) if ws is not None else None, B, T, h, TSG_F32, st)
    torch.cuda.synchronize(); assert rc == 0
    return out
ref = run(None)
for i in range(reps):
    ws = torch.zeros(512, dtype=torch.int32, device="cuda")
    out = run(ws)
    d = (out - ref).abs()
    bad_t = (d.amax(dim=(1, 2)) > 1e-5).nonzero().flatten().tolist()
    print("ws[0:4]", ws[:4].tolist())
    print(f"mode {os.environ.get('TSG_LSTM_PERSIST')} B={B} T={T} h={h} rep {i}: err flag {int(ws[0])}  max diff {d.max().item():.3e}  nan {int(torch.isnan(out).sum())}  first bad t: {bad_t[:6]}")
    if d.max() > 1e-5:
        for t in (0, 1, T - 1):
            f, r = d[t, :, :h], d[t, :, h:]
            print(f"   t={t}: fwd-half max {f.max().item():.3e} bad rows {(f.amax(1) > 1e-5).sum().item()}/{B} bad cols {(f.amax(0) > 1e-5).sum().item()}/{h} | rev-half max {r.max().item():.3e} bad rows {(r.amax(1) > 1e-5).sum().item()} bad cols {(r.amax(0) > 1e-5).sum().item()}")
        print("   sample out vs ref at t=1, b=0, u=0..7:", out[1, 0, :8].tolist(), ref[1, 0, :8].tolist())
